# round 3, call K: split walk records + tie-break bit: GPU suite, timelines, hole fuzz sample, bench
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" > gpurun_out/r03k_gpu_tests.txt; tail -12 gpurun_out/r03k_gpu_tests.txt
cd /tmp
for cfg in "15000 0 frame c2frame" "45000 2 frame c4frame" "150000 3 1048576 c5rays"; do
  set -- $cfg
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $1 $2 $3 > /tmp/log.txt 2>&1
  ( grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -8 ) > $GRAFT_REPO_ROOT/gpurun_out/r03k_$4_timeline.txt
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/r03k_*_timeline.txt
timeout 300 python profiles/r03_hole_fuzz.py 30 0 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 --no-render > gpurun_out/r03k_bench.json 2> gpurun_out/r03k_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03k_bench.json'))
print('headline', d['ms_per_step'], d['roofline']['frac'], d['walk_hand_over_reasons'])
for k,v in d['configs'].items():
    if 'ms' in v: print(k, round(v['ms'],3), round(v['roofline']['frac'],3))
PY
