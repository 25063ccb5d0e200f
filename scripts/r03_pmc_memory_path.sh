# Round 3: what the vector-memory path does in the frames whose tree is read from global memory (TA = address unit, TCP = per-CU L1).
# Two counters per pass (the TA block has few slots: five at once abort rocprofv3 with "exceeds the capabilities of the hardware"), each pass
# under its own timeout.
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_mem
mkdir -p $O
pass() { wl=$1; name=$2; shift 2; timeout -k 10 170 rocprofv3 --pmc "$@" --output-format csv -d $O/${wl}_$name -- python3 $ARGS > $O/${wl}_$name.log 2>&1; echo "$wl $name rc=$?"; }
for wl in million teapot; do
  spp=64; [ $wl = teapot ] && spp=256
  ARGS="bench.py --steps 1 --warmup 0 --spp $spp --workload $wl --no-cpu-baseline --no-f32-line --no-optin-line --no-reference-line"
  pass $wl ta1 TA_TA_BUSY_sum GRBM_GUI_ACTIVE || exit 1
  pass $wl ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
  pass $wl tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_PERF_SEL_TOTAL_READ
  pass $wl tcp2 TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_TAGRAM0_REQ_sum
done
python3 scripts/summarize_pmc.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
