set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r01f
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01f/kt -- python3 $R/bench.py --steps 3 --no-cpu-baseline > $R/gpurun_out/r01f/kt_bench.json 2> $R/gpurun_out/r01f/kt.err )
echo "kernel-trace done"
python bench.py > gpurun_out/r01f/bench.json 2> gpurun_out/r01f/bench.err; echo head done
python bench.py --real f64 --no-cpu-baseline > gpurun_out/r01f/bench_f64.json 2>/dev/null; echo f64 done
python bench.py --workload teapot --no-cpu-baseline --steps 2 > gpurun_out/r01f/bench_teapot.json 2>/dev/null; echo teapot done
python bench.py --workload million --no-cpu-baseline --steps 2 > gpurun_out/r01f/bench_million.json 2>/dev/null; echo million done
python bench.py --workload movie --no-cpu-baseline --steps 3 > gpurun_out/r01f/bench_movie.json 2>/dev/null; echo movie done
for b in sah ordered; do python bench.py --bvh $b --no-cpu-baseline --steps 3 > gpurun_out/r01f/bench_$b.json 2>/dev/null; echo $b done; done
python bench.py --workload teapot --bvh sah --no-cpu-baseline --steps 2 > gpurun_out/r01f/bench_teapot_sah.json 2>/dev/null
python bench.py --workload million --bvh sah --no-cpu-baseline --steps 2 > gpurun_out/r01f/bench_million_sah.json 2>/dev/null; echo sah extras done
python bench.py --bvh lbvh --no-cpu-baseline --steps 3 > gpurun_out/r01f/bench_lbvh.json 2>/dev/null
python bench.py --workload million --bvh lbvh --no-cpu-baseline --steps 2 > gpurun_out/r01f/bench_million_lbvh.json 2>/dev/null; echo lbvh done
SPP=512 bash scripts/profile_pmc.sh
