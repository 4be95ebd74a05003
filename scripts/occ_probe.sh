for b in 1 2 3 4; do
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --opt 9=$b > gpurun_out/occ_$b.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/occ_$b.json'));print($b, d['value'], d['roofline']['kernel_ms'])"
done
