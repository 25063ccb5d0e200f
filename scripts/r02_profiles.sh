# Round-2 measurement set (one gpurun call): bench lines, kernel trace, PMC passes -> gpurun_out/r02p/, then profiles/
set -u
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r02p
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python3 $R/bench.py --steps 3 --no-cpu-baseline --no-f32-line --no-optin-line > $R/$O/kt_bench.json 2> $R/$O/kt.err ); echo "kernel-trace rc=$?"
python bench.py --real f32 --no-cpu-baseline > $O/bench_f32.json 2>/dev/null; echo f32 done
python bench.py --workload teapot --no-cpu-baseline --steps 2 > $O/bench_teapot.json 2>/dev/null; echo teapot done
python bench.py --workload million --no-cpu-baseline --steps 2 > $O/bench_million.json 2>/dev/null; echo million done
python bench.py --workload movie --no-cpu-baseline --steps 3 > $O/bench_movie.json 2>/dev/null; echo movie done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --spp 64 --reduce gloo --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2-rank rehearsal rc=$?"
SPP=512 REAL=f64 WORKLOAD=book1 TAG=book1_f64 bash scripts/profile_pmc.sh
SPP=1024 REAL=f64 WORKLOAD=teapot TAG=teapot_f64 bash scripts/profile_pmc.sh
SPP=256 REAL=f64 WORKLOAD=million TAG=million_f64 bash scripts/profile_pmc.sh
SPP=512 REAL=f32 WORKLOAD=book1 TAG=book1_f32 bash scripts/profile_pmc.sh
