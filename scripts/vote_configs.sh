#!/bin/bash
# On the GPU box: the other bench configurations under two settings of the vote weights.
mkdir -p gpurun_out/votecfg
for cfg in c2 c5 c3; do
for w in "$@"; do
  IFS=, read a b c <<< "$w"
  v=$((a | (b << 8) | (c << 16)))
  timeout -k 10 300 python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extras --opt 8=$v > gpurun_out/votecfg/bench_${cfg}_$w.json 2> gpurun_out/votecfg/bench_${cfg}_$w.err || { echo "bench $cfg $w failed"; tail -5 gpurun_out/votecfg/bench_${cfg}_$w.err; exit 1; }
  python3 - gpurun_out/votecfg/bench_${cfg}_$w.json $cfg $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('%s weights %s: Mrays/s %.0f  kernel_ms %.3f' % (sys.argv[2], sys.argv[3], d['value'], r['kernel_ms']))
PY
done
done
