#!/bin/bash
# On the GPU box: the bench workload on the binned builder's trees with other leaf sizes.  Arguments: leaf sizes.
mkdir -p gpurun_out/leaf
for n in "$@"; do
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --leaf-max $n > gpurun_out/leaf/bench_$n.json 2> gpurun_out/leaf/bench_$n.err || { echo "bench $n failed"; tail -n 5 gpurun_out/leaf/bench_$n.err; exit 1; }
  python3 - gpurun_out/leaf/bench_$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; s=r['schedule_per_launch']; c=r['counters_per_launch']
print('leaf_max %s: Mrays/s %.0f  kernel_ms %.3f  lanes/iter %s Miters %s box %.2fG tri %.0fM' % (sys.argv[2], d['value'], r['kernel_ms'], [round(l/max(1,i),1) for l,i in zip(s['lanes'],s['iterations'])], [round(i/1e6,2) for i in s['iterations']], c['box_tests']/1e9, c['triangle_tests']/1e6))
PY
done
