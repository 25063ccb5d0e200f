# Round-3 measurement set (one gpurun call): bench lines, kernel trace, PMC passes -> gpurun_out/r03p/, then profiles/ (scripts/r03_collect.py)
set -u
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r03p
mkdir -p $O
if [ "${PART:-1}" = 1 ]; then
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python3 $R/bench.py --steps 3 --no-cpu-baseline --no-f32-line --no-optin-line --no-reference-line > $R/$O/kt_bench.json 2> $R/$O/kt.err ); echo "kernel-trace rc=$?"
python bench.py --sum-order reference --no-cpu-baseline --no-f32-line --no-optin-line > $O/bench_reference_order.json 2>/dev/null; echo reference-order done
python bench.py --real f32 --no-cpu-baseline > $O/bench_f32.json 2>/dev/null; echo f32 done
python bench.py --workload teapot --no-cpu-baseline --steps 2 > $O/bench_teapot.json 2>/dev/null; echo teapot done
python bench.py --workload million --no-cpu-baseline --steps 2 > $O/bench_million.json 2>/dev/null; echo million done
python bench.py --workload movie --no-cpu-baseline --steps 3 > $O/bench_movie.json 2>/dev/null; echo movie done
for n in 256 128 64; do python bench.py --spp $n --no-cpu-baseline --no-f32-line --no-optin-line --no-reference-line --steps 5 > $O/bench_shard_$n.json 2>/dev/null; done; echo shards done
fi
if [ "${PART:-1}" = 2 ]; then
SPP=512 REAL=f64 WORKLOAD=book1 TAG=book1_f64 bash scripts/profile_pmc.sh
SPP=1024 REAL=f64 WORKLOAD=teapot TAG=teapot_f64 bash scripts/profile_pmc.sh
SPP=256 REAL=f64 WORKLOAD=million TAG=million_f64 bash scripts/profile_pmc.sh
SPP=512 REAL=f64 WORKLOAD=movie TAG=movie_f64 bash scripts/profile_pmc.sh
SPP=512 REAL=f32 WORKLOAD=book1 TAG=book1_f32 bash scripts/profile_pmc.sh
fi
