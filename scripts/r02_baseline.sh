set -u
mkdir -p gpurun_out/r02a
python bench.py --real f64 --steps 3 > gpurun_out/r02a/bench_f64.json 2> gpurun_out/r02a/bench_f64.err; echo "bench f64 rc=$?"
python scripts/gpu_sweep_inproc.py book1 f64 1920 64 "10,56;6,56;16,56;0,56;10,48;10,60;10,64;16,60;16,64;24,60" > gpurun_out/r02a/sweep_f64.txt 2>&1; echo "sweep rc=$?"
python scripts/gpu_sweep_inproc.py million f64 3840 16 "10,56;16,56;10,48" > gpurun_out/r02a/sweep_million_f64.txt 2>&1; echo "sweep million rc=$?"
SPP=512 REAL=f64 WORKLOAD=book1 TAG=book1_f64 bash scripts/profile_pmc.sh
SPP=128 REAL=f64 WORKLOAD=teapot TAG=teapot_f64 bash scripts/profile_pmc.sh
SPP=32 REAL=f64 WORKLOAD=million TAG=million_f64 bash scripts/profile_pmc.sh
SPP=32 REAL=f32 WORKLOAD=million TAG=million_f32 bash scripts/profile_pmc.sh
