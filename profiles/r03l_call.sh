# round 3, call L: where a training iteration goes (kernel stats), counters of the stand-alone gather
set -x
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p4; rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > $GRAFT_REPO_ROOT/gpurun_out/r03l_train_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r03l_train_kernel_stats.txt
rm -rf /tmp/p5; rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d /tmp/p5 -o g -- python $GRAFT_REPO_ROOT/profiles/r02_gather_match_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r03l_gather_bench.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p5 -name "*.db" | head -1) | grep -E "^#|^kernel|k_interp|k_find_matched|k_transpose" > $GRAFT_REPO_ROOT/gpurun_out/r03l_gather_pmc.txt
rm -rf /tmp/p6; rocprofv3 --kernel-trace --stats -d /tmp/p6 -o g -- python $GRAFT_REPO_ROOT/profiles/r02_gather_match_bench.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p6 -name "*.db" | head -1) | grep -E "^#|^kernel|k_interp|k_find_matched|k_transpose" > $GRAFT_REPO_ROOT/gpurun_out/r03l_gather_kernel_stats.txt
cd $GRAFT_REPO_ROOT; grep -v "^W2026\|^E2026" gpurun_out/r03l_train_once.txt | tail -2; head -30 gpurun_out/r03l_train_kernel_stats.txt | cut -c1-80,112-160; cat gpurun_out/r03l_gather_pmc.txt | cut -c1-60,82-150; cat gpurun_out/r03l_gather_kernel_stats.txt | cut -c1-70,112-160; grep -v "^W2026" gpurun_out/r03l_gather_bench.txt | tail -9
