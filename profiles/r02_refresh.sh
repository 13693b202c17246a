# Round-2 evidence refresh (run through gpurun from the repository root): GPU suite, smoke, the bench line, the
# rocprofv3 kernel-trace summary and the separate PMC passes of the SAME trace-only bench command, the training
# adjoint bench + its kernel trace, gather/match, small batches, the writer-grid sweep.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -14 > gpurun_out/r02_gpu_tests.txt; cat gpurun_out/r02_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r02_smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -c 400 gpurun_out/r02_bench.json
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render --no-configs"
B3="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-configs"
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- $B > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_kernel_stats.txt
rm -rf /tmp/p2; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p2 -o w -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_pmc_write.txt
rm -rf /tmp/p3; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o f -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p3 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_pmc_fetch.txt
rm -rf /tmp/p4; rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r02_train_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r02_train_bench.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02_train_kernel_stats.txt
cd $GRAFT_REPO_ROOT
python profiles/r02_gather_match_bench.py > gpurun_out/r02_gather_match.txt 2>&1
python profiles/r02_small_batch.py > gpurun_out/r02_small_batch.txt 2>&1
python profiles/r02_walk_sweep.py c2 c4 c5 > gpurun_out/r02_walk_sweep.txt 2>&1
head -9 gpurun_out/r02_trace_kernel_stats.txt | cut -c1-160; cat gpurun_out/r02_trace_pmc_write.txt gpurun_out/r02_trace_pmc_fetch.txt | cut -c1-180
grep -v "^W2026\|^E2026" gpurun_out/r02_train_bench.txt | tail -3; cat gpurun_out/r02_walk_sweep.txt | head -12
