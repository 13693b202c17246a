# round 4, final call (after the dense_tails fix and the compact rows on the BVH path):  GPU suite (plain + poisoned buffers), smoke, bench line, per-config timelines,
# kernel stats + PMC traffic of the bench command, kernel stats of whole training iterations
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" > gpurun_out/r04final_gpu_tests.txt; tail -8 gpurun_out/r04final_gpu_tests.txt
TETRANERF_HIP_POISON=1 timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "^W2026" > gpurun_out/r04final_gpu_tests_poison.txt; tail -3 gpurun_out/r04final_gpu_tests_poison.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04final_bench.json 2> gpurun_out/r04final_bench.err; tail -c 300 gpurun_out/r04final_bench.json
cd /tmp
for cfg in "15000 0 frame c2frame" "45000 2 frame c4frame" "150000 3 1048576 c5rays"; do
  set -- $cfg
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $1 $2 $3 > /tmp/log.txt 2>&1
  ( grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -10 ) > $GRAFT_REPO_ROOT/gpurun_out/r04final_$4_timeline.txt
done
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render --no-configs"
B3="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-configs"
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- $B > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04final_trace_kernel_stats.txt
rm -rf /tmp/p2; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p2 -o w -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04final_trace_pmc_write.txt
rm -rf /tmp/p3; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o f -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p3 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04final_trace_pmc_fetch.txt
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > $GRAFT_REPO_ROOT/gpurun_out/r04final_train_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04final_train_kernel_stats.txt
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03ae_train_tn.py > $GRAFT_REPO_ROOT/gpurun_out/r04final_train_tn_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04final_train_tn_kernel_stats.txt
cd $GRAFT_REPO_ROOT; cat gpurun_out/r04final_*_timeline.txt; head -9 gpurun_out/r04final_trace_kernel_stats.txt | cut -c1-60,112-160
grep -E "k_trace_walk|k_write_seg|k_fill|k_postprocess_log|k_trace_general|k_verify" gpurun_out/r04final_trace_pmc_write.txt gpurun_out/r04final_trace_pmc_fetch.txt | cut -c1-90,110-200
