cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_train_gpu.py tests/test_reference_model_gpu.py tests/test_plugin_gpu.py -q -x -m gpu 2>&1 | tail -3
cd /tmp
rm -rf /tmp/p4; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > /tmp/o.txt 2>&1
echo "== hoisted gather: $(grep 'ms per iteration' /tmp/o.txt)"
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) | grep -E "k_dw_gemm|k_mlp_forward|k_mlp_backward" | cut -c1-70,112-160
