#!/bin/bash
# usage: leaf_sweep.sh <builder> <leaf sizes...>
b=$1; shift
for l in "$@"; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --builder $b --leaf-max $l > gpurun_out/leaf_${b}_$l.json 2>/dev/null || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/leaf_${b}_$l.json'));c=d['roofline']['counters_per_launch']
print('$b leaf $l:', round(d['value']), 'Mrays/s', round(d['roofline']['kernel_ms'],2), 'ms; per ray: internal', round(c['internal_visits']/c['rays'],1), 'tri tests', round(c['triangle_tests']/c['rays'],1))"
done
