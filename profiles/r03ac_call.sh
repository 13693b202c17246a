# round 3, call AC: gather with any-slot row reuse: parity tests of the gather + stand-alone timing (kernel stats)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -k "interp or gather or reference_suite or mlp_forward" > gpurun_out/r03ac_tests.txt 2>&1; tail -3 gpurun_out/r03ac_tests.txt
cd /tmp; rm -rf /tmp/p6; rocprofv3 --kernel-trace --stats -d /tmp/p6 -o g -- python $GRAFT_REPO_ROOT/profiles/r02_gather_match_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r03ac_gather_bench.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p6 -name "*.db" | head -1) | grep -E "^#|^kernel|k_interp|k_find_matched" | cut -c1-70,112-160; grep "interpolate_values fwd" $GRAFT_REPO_ROOT/gpurun_out/r03ac_gather_bench.txt
