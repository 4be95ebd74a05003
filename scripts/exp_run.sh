#!/bin/bash
# on the GPU box: bench each named experimental build, print value and kernel time
for n in "$@"; do
  timeout -k 10 150 python scripts/exp_bench.py scripts/_build/librt_$n.so --steps 8 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/exp_$n.json 2> gpurun_out/exp_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/exp_$n.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/exp_$n.json'));s=d['roofline']['schedule_per_launch']
print('$n', round(d['value']), round(d['roofline']['kernel_ms'],3), 'lanes/iter', [round(l/max(1,i),1) for l,i in zip(s['lanes'],s['iterations'])], 'Miters', [round(i/1e6,2) for i in s['iterations']])"
done
