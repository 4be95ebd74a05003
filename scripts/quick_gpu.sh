#!/bin/bash
# On the GPU box: the parity tests that exercise the kernels, then a short bench of the in-tree library
# (timing) and one counter pass (vector / scalar instructions per launch).  Output under gpurun_out/$1.
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests/test_gpu_scenes.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > "$OUT/tests.log" 2>&1
echo "tests exit $?" | tee -a "$OUT/tests.log"
tail -3 "$OUT/tests.log"
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo bench failed; tail -5 "$OUT/bench.err"; exit 1; }
python3 - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; s=r['schedule_per_launch']
print('Mrays/s %.0f  kernel_ms %.3f  clock %.0f' % (d['value'], r['kernel_ms'], r['kernel_clock_mhz']))
print('lanes/iter', [round(l/max(1,i),1) for l,i in zip(s['lanes'],s['iterations'])], 'Miters', [round(i/1e6,2) for i in s['iterations']])
PY
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/pmc" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_pmc.json" 2> "$OUT/pmc.err" || { echo pmc failed; tail -3 "$OUT/pmc.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/pmc/*/*counter_collection.csv')[0]
per=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if 'path_kernel_persistent<32, false, false>' in r['Kernel_Name']:
        per[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
n=len(per); tot=collections.defaultdict(float)
for v in per.values():
    for k,x in v.items(): tot[k]+=x
print('launches', n, {k: round(x/n/1e9,3) for k,x in tot.items()}, 'G per launch')
PY
