#!/bin/bash
# On the GPU box: the one-frame configuration (c2) under fewer resident workgroups per CU.
mkdir -p gpurun_out/c2b
for b in 4 3 2 4 3; do
  timeout -k 10 200 python3 bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --opt 9=$b > gpurun_out/c2b/b$b.json 2> gpurun_out/c2b/b$b.err || { echo "failed $b"; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/c2b/b$b.json').read().strip().splitlines()[-1]);print('blocks/CU $b', round(d['value']), d['roofline']['kernel_ms'])"
done
