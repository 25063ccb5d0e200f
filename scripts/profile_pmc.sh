#!/bin/bash
# PMC passes for the render kernel (run on the GPU box through gpurun).  Counters are collected
# in their own runs (no --kernel-trace/--stats), a few per pass (SQ 8 slots, TCC 4: FETCH_SIZE=3, WRITE_SIZE=2).
#   SPP=512 REAL=f64 WORKLOAD=book1 TAG=book1_f64 bash scripts/profile_pmc.sh   -> gpurun_out/pmc_$TAG/
set -u
export TMPDIR=/tmp
SPP=${SPP:-64}
REAL=${REAL:-f64}
WORKLOAD=${WORKLOAD:-book1}
TAG=${TAG:-${WORKLOAD}_${REAL}}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
ARGS="bench.py --steps 1 --warmup 0 --spp $SPP --real $REAL --workload $WORKLOAD --no-cpu-baseline --no-f32-line --no-optin-line --no-reference-line ${EXTRA_ARGS:-}"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1; echo "pass $TAG/$name rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass sq3 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum
python3 scripts/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
