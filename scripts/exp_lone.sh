#!/bin/bash
# on the GPU box: lone-frame and fused timings of experimental builds
for n in "$@"; do
  timeout -k 10 200 python scripts/exp_bench.py scripts/_build/librt_$n.so --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/exp_$n.json 2> gpurun_out/exp_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/exp_$n.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/exp_$n.json'))
print('$n', 'fused', round(d['value']), round(d['ms_per_step'],2), 'lone frame ms', round(d['ms_per_lone_frame'],3), 'first', round(d['first_launch_of_a_view']['kernel_ms'],2), 'floor', round(d['strong_scaling_floor']['ms_per_step'],2))"
done
