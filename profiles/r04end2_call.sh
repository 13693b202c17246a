# round 4, after the closing call: GPU suite with poisoned buffers on the final code + PMC traffic of the bench command
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
TETRANERF_HIP_POISON=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^W2026" > gpurun_out/r04end_gpu_tests_poison.txt; tail -3 gpurun_out/r04end_gpu_tests_poison.txt
cd /tmp
B3="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-configs"
rm -rf /tmp/p2; timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p2 -o w -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04end_trace_pmc_write.txt
rm -rf /tmp/p3; timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o f -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p3 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04end_trace_pmc_fetch.txt
cd $GRAFT_REPO_ROOT
grep -E "k_trace_walk|k_write_seg|k_fill|k_postprocess_log|k_trace_general|k_verify" gpurun_out/r04end_trace_pmc_write.txt gpurun_out/r04end_trace_pmc_fetch.txt | cut -c1-90,110-220
