#!/bin/bash
# On the GPU box: experimental builds of the library (scripts/build_variant.sh NAME ...) benched in turn,
# three rounds, on this one box: kernel ms per launch of the bench workload for each.
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for n in "$@"; do
    timeout -k 10 150 python scripts/exp_bench.py scripts/_build/librt_$n.so --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/ab/$n.$r.json 2> gpurun_out/ab/$n.$r.err || { echo "$n FAILED"; tail -3 gpurun_out/ab/$n.$r.err; exit 1; }
  done
done
python3 - "$@" <<'PY'
import json,sys
for n in sys.argv[1:]:
    ms=[json.loads(open(f'gpurun_out/ab/{n}.{r}.json').read().strip().splitlines()[-1])['roofline']['kernel_ms'] for r in (1,2,3)]
    print(n, ' '.join(f'{m:.3f}' for m in ms), 'mean %.3f' % (sum(ms)/3))
PY
