# Round-2b evidence refresh (run through gpurun from the repository root): GPU suite, smoke, the bench line, the
# rocprofv3 kernel-trace summary and the separate PMC passes of the SAME trace-only bench command, per-config timelines,
# the training adjoint bench, gather/match, small batches.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v "^W2026" | grep -E "passed|failed|error|^[0-9.]+s " | tail -10 > gpurun_out/r02z_gpu_tests.txt; cat gpurun_out/r02z_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r02z_smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02z_bench.json 2> gpurun_out/r02z_bench.err; tail -c 300 gpurun_out/r02z_bench.json
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render --no-configs"
B3="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-configs"
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- $B > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02z_trace_kernel_stats.txt
rm -rf /tmp/p2; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p2 -o w -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02z_trace_pmc_write.txt
rm -rf /tmp/p3; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o f -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p3 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02z_trace_pmc_fetch.txt
for cfg in "15000 0 frame c2frame" "45000 2 frame c4frame" "150000 3 frame c5frame" "150000 3 1048576 c5rays"; do
  set -- $cfg
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $1 $2 $3 > /tmp/log.txt 2>&1
  ( grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -7 ) > $GRAFT_REPO_ROOT/gpurun_out/r02z_$4_timeline.txt
done
rm -rf /tmp/p4; rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r02_train_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r02z_train_bench.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02z_train_kernel_stats.txt
cd $GRAFT_REPO_ROOT
python profiles/r02_gather_match_bench.py 2>&1 | grep -v "^W2026" > gpurun_out/r02z_gather_match.txt
head -9 gpurun_out/r02z_trace_kernel_stats.txt | cut -c1-60,112-160; cat gpurun_out/r02z_trace_pmc_write.txt gpurun_out/r02z_trace_pmc_fetch.txt | cut -c1-60,82-180
grep -v "^W2026\|^E2026" gpurun_out/r02z_train_bench.txt | tail -3; cat gpurun_out/r02z_c4frame_timeline.txt
