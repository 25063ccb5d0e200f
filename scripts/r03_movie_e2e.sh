# Round 3, VERDICT item 4(b): all 240 frames of configs[4] on one GPU with P3, P6 and PNG output (bench.py --end-to-end 240)
mkdir -p gpurun_out/r03p
python bench.py --workload movie --no-cpu-baseline --steps 3 --end-to-end ${FRAMES:-240} > gpurun_out/r03p/bench_movie_e2e.json 2> gpurun_out/r03p/bench_movie_e2e.err; echo "movie end-to-end rc=$?"
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r03p/bench_movie_e2e.json') if l.startswith('{')][-1]); print(json.dumps(d['end_to_end'], indent=1))"
