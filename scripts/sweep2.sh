# usage: sweep2.sh <pipeline> "<opts>" ...   each opts = space-separated ID=VALUE
p=$1; shift
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --pipeline $p $args > gpurun_out/sw.json 2>/dev/null || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/sw.json'));s=d['roofline']['schedule_per_launch']
tot=sum(s['cycles'])+s['shade'][2]
print('$o', round(d['value']), 'util', [round(l/max(1,i)/64,2) for l,i in zip(s['lanes'],s['iterations'])], 'share', [round(c/tot,2) for c in s['cycles']], 'shade', round(s['shade'][2]/tot,2), round(s['shade'][1]/max(1,s['shade'][0])/64,2), 'Gcyc', round(tot/1e9))"
done
