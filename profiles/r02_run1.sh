# round 2, first GPU call: parity of the log-walk + row-writer kernels, writer-grid sweep, kernel trace of the bench
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_walk_gpu.py tests/test_trace_gpu.py tests/test_parity_configs_gpu.py -x -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/r02_run1_tests.txt
cat gpurun_out/r02_run1_tests.txt
timeout 300 python profiles/r02_walk_sweep.py c2 c4 > gpurun_out/r02_walk_sweep.txt 2>&1
cat gpurun_out/r02_walk_sweep.txt
cd /tmp; rm -rf /tmp/p1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render > $GRAFT_REPO_ROOT/gpurun_out/r02_prof1_bench.log 2>&1
tail -c 1500 $GRAFT_REPO_ROOT/gpurun_out/r02_prof1_bench.log
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats.txt
head -24 $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats.txt
