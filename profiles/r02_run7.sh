set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r02_run7_tests.txt
cat gpurun_out/r02_run7_tests.txt
timeout 300 python profiles/r02_walk_sweep.py c2 c4 c5 > gpurun_out/r02_walk_sweep7.txt 2>&1
cat gpurun_out/r02_walk_sweep7.txt
timeout 300 python profiles/r02_gather_match_bench.py > gpurun_out/r02_gather_match.txt 2>&1
cat gpurun_out/r02_gather_match.txt
cd /tmp; rm -rf /tmp/p1
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render --no-configs"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- $B > $GRAFT_REPO_ROOT/gpurun_out/r02_prof7_bench.log 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats.txt
head -12 $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats.txt | cut -c1-160
grep -o '"ms_per_step": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/r02_prof7_bench.log
