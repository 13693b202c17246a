set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_walk_gpu.py tests/test_parity_configs_gpu.py tests/test_trace_gpu.py -m gpu -q 2>&1 | grep -v "^W2026" | tail -5 > gpurun_out/r06q_tests.txt; cat gpurun_out/r06q_tests.txt
timeout 1200 python profiles/r06e_sweep.py 7 C2,C4,C5 auto: pairwise:cert_ends=0 pairwise_abc:cert_ends=3 clusters:cert_ends=1 2>&1 | grep -v amdgpu.ids > gpurun_out/r06q_order_sweep.txt; cut -c1-150 gpurun_out/r06q_order_sweep.txt
