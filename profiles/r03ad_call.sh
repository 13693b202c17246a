# round 3, call AD: gather adjoint that carries the sums of the vertices two consecutive tetrahedra share: parity + timing
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -k "interp or gather or reference_suite or train or plugin" > gpurun_out/r03ad_tests.txt 2>&1; tail -2 gpurun_out/r03ad_tests.txt
cd /tmp; rm -rf /tmp/p6; rocprofv3 --kernel-trace --stats -d /tmp/p6 -o g -- python $GRAFT_REPO_ROOT/profiles/r02_gather_match_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r03ad_gather_bench.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p6 -name "*.db" | head -1) | grep -E "k_interp_bwd" | cut -c1-70,112-160
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03ae_train_tn.py > /tmp/log.txt 2>&1; grep "ms per" /tmp/log.txt; python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) | grep "k_interp_bwd" | cut -c1-70,112-160
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > /tmp/log.txt 2>&1; grep "ms per" /tmp/log.txt; python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) | grep "k_interp_bwd" | cut -c1-70,112-160
