#!/bin/bash
# usage: weights_sweep.sh c,b,l ...   (vote weights control:box:leaf of the persistent kernel)
for w in "$@"; do
  IFS=, read c b l <<< "$w"
  v=$((c | (b<<8) | (l<<16)))
  timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --opt 8=$v > gpurun_out/sw.json 2>/dev/null || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/sw.json'));s=d['roofline']['schedule_per_launch']
print('$w', round(d['value']), round(d['roofline']['kernel_ms'],2), 'lanes', [round(l/max(1,i),1) for l,i in zip(s['lanes'],s['iterations'])], 'Miters', [round(i/1e6,2) for i in s['iterations']])"
done
