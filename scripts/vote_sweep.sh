#!/bin/bash
# On the GPU box: the bench workload under a few settings of the vote weights (RT_OPT_VOTE_WEIGHTS = 8,
# control | box << 8 | leaf << 16).  Arguments: triples like 3,2,6.  Output under gpurun_out/vote.
mkdir -p gpurun_out/vote
for w in "$@"; do
  IFS=, read a b c <<< "$w"
  v=$((a | (b << 8) | (c << 16)))
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --opt 8=$v > gpurun_out/vote/bench_$w.json 2> gpurun_out/vote/bench_$w.err || { echo "bench $w failed"; tail -5 gpurun_out/vote/bench_$w.err; exit 1; }
  python3 - gpurun_out/vote/bench_$w.json $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; s=r['schedule_per_launch']
print('weights %s: Mrays/s %.0f  kernel_ms %.3f  lanes/iter %s Miters %s' % (sys.argv[2], d['value'], r['kernel_ms'], [round(l/max(1,i),1) for l,i in zip(s['lanes'],s['iterations'])], [round(i/1e6,2) for i in s['iterations']]))
PY
done
