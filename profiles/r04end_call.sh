# round 4, closing call (after the adaptive sort of the logged hits and the batched gathers of the pairing kernels):
# GPU suite, smoke, bench line, per-config timelines, kernel stats of the bench command
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" > gpurun_out/r04end_gpu_tests.txt; tail -8 gpurun_out/r04end_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04end_bench.json 2> gpurun_out/r04end_bench.err; tail -c 300 gpurun_out/r04end_bench.json
cd /tmp
for cfg in "15000 0 frame c2frame" "45000 2 frame c4frame" "150000 3 1048576 c5rays"; do
  set -- $cfg
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $1 $2 $3 > /tmp/log.txt 2>&1
  ( grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -10 ) > $GRAFT_REPO_ROOT/gpurun_out/r04end_$4_timeline.txt
done
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render --no-configs"
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- $B > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04end_trace_kernel_stats.txt
cd $GRAFT_REPO_ROOT; cat gpurun_out/r04end_*_timeline.txt; head -9 gpurun_out/r04end_trace_kernel_stats.txt | cut -c1-60,112-160
