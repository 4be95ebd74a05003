# usage: sweep.sh <pipeline> <weights...>   (weights as c,b,l)
p=$1; shift
for w in "$@"; do
  IFS=, read c b l <<< "$w"
  v=$((c | (b<<8) | (l<<16)))
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --pipeline $p --opt 8=$v > gpurun_out/sw.json 2>/dev/null || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/sw.json'));s=d['roofline']['schedule_per_launch']
tot=sum(s['cycles'])+s['shade'][2]
print('$w', round(d['value']), 'util', [round(l/max(1,i)/64,2) for l,i in zip(s['lanes'],s['iterations'])], 'share', [round(c/tot,2) for c in s['cycles']], 'shade', round(s['shade'][2]/tot,2), round(s['shade'][1]/max(1,s['shade'][0])/64,2), s['shade'][0])"
done
