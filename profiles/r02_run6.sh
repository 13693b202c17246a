set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 > gpurun_out/r02_run6_tests.txt
cat gpurun_out/r02_run6_tests.txt
timeout 300 python profiles/r02_gather_match_bench.py > gpurun_out/r02_gather_match.txt 2>&1
cat gpurun_out/r02_gather_match.txt
cd /tmp; rm -rf /tmp/p1 /tmp/p2 /tmp/p3
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render --no-configs"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- $B > $GRAFT_REPO_ROOT/gpurun_out/r02_prof6_bench.log 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats.txt
head -12 $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats.txt | cut -c1-160
grep -o '"ms_per_step": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/r02_prof6_bench.log
B3="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-configs"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p2 -o w -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_pmc_write.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o f -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p3 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_pmc_fetch.txt
cat $GRAFT_REPO_ROOT/gpurun_out/r02_trace_pmc_write.txt $GRAFT_REPO_ROOT/gpurun_out/r02_trace_pmc_fetch.txt | cut -c1-200
cd $GRAFT_REPO_ROOT; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_full.json 2> gpurun_out/r02_bench_full.err; tail -c 3000 gpurun_out/r02_bench_full.json; tail -3 gpurun_out/r02_bench_full.err
