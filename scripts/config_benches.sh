#!/bin/bash
# on the GPU box: one bench line per benchmark configuration (c2, c5, c3 with leaves of 16, c4), into gpurun_out/cfg_<name>.json
set -u
run() { # name, args...
  local n=$1; shift
  timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/cfg_$n.json 2> gpurun_out/cfg_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/cfg_$n.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/cfg_$n.json'));print('$n', round(d['value']), round(d['ms_per_step'],3))"
}
run c2 --config c2 --steps 20 --warmup 3 &&
run c5 --config c5 --steps 10 --warmup 2 &&
run c3_leaf16 --config c3 --leaf-max 16 --steps 10 --warmup 2 &&
run c4 --config c4 --steps 3 --warmup 1
