# round 4, call K: kernel breakdown of whole training iterations (sync-free forward), both shipped configurations
set -x
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03ae_train_tn.py > $GRAFT_REPO_ROOT/gpurun_out/r04k_train_tn_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04k_train_tn_kernel_stats.txt
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > $GRAFT_REPO_ROOT/gpurun_out/r04k_train_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04k_train_kernel_stats.txt
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/r04k_train_tn_once.txt; head -45 gpurun_out/r04k_train_tn_kernel_stats.txt | cut -c1-70,112-160; tail -1 gpurun_out/r04k_train_once.txt
