# opt-in trees in f64 at BASELINE sizes -> gpurun_out/r03o/
O=gpurun_out/r03o; mkdir -p $O
for b in sah ordered lbvh; do
  python bench.py --bvh $b --no-cpu-baseline --no-f32-line --no-reference-line --steps 3 > $O/bench_$b.json 2>/dev/null; echo "book1 $b rc=$?"
done
for b in sah ordered lbvh; do
  python bench.py --workload teapot --bvh $b --no-cpu-baseline --no-f32-line --no-reference-line --steps 2 > $O/bench_teapot_$b.json 2>/dev/null; echo "teapot $b rc=$?"
  python bench.py --workload million --bvh $b --no-cpu-baseline --no-f32-line --no-reference-line --steps 2 > $O/bench_million_$b.json 2>/dev/null; echo "million $b rc=$?"
done
python scripts/build_time.py > $O/build_time.txt 2>&1
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); c=d['roofline']['counters_per_launch']
print('$f'.split('/')[-1], d['value'], 'node/seg %.1f prim/seg %.2f'%(c['node_tests']/c['segments'], c['prim_tests']/c['segments']))"; done
cat $O/build_time.txt | tail -8
