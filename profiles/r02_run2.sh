set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python profiles/r02_debug_parity.py lattice_jittered > gpurun_out/r02_debug_lattice.txt 2>&1
head -60 gpurun_out/r02_debug_lattice.txt
timeout 1200 python -m pytest tests/test_walk_gpu.py tests/test_trace_gpu.py tests/test_parity_configs_gpu.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r02_run2_tests.txt
cat gpurun_out/r02_run2_tests.txt
timeout 300 python profiles/r02_walk_sweep.py c2 c4 > gpurun_out/r02_walk_sweep2.txt 2>&1
cat gpurun_out/r02_walk_sweep2.txt
cd /tmp; rm -rf /tmp/p1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render > $GRAFT_REPO_ROOT/gpurun_out/r02_prof2_bench.log 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats2.txt
head -12 $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats2.txt | cut -c1-160
