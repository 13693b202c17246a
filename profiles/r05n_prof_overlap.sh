cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p6; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p6 -o t -- python $GRAFT_REPO_ROOT/profiles/r03ae_train_tn.py > /tmp/train.log 2>&1
( grep "ms per iteration" /tmp/train.log; python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p6 -name "*.db" | head -1) ) | head -16 | cut -c1-70,112-160
