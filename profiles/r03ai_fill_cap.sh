for cap in 512 256 384 768 1024 2048 512; do
  echo -n "fill cap $cap: "; TN_EXP_FILL_CAP=$cap timeout 300 python bench.py --steps 20 --warmup 5 --no-render --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],3), round(j['roofline']['frac'],4))"
done
