# round 6: OrderR6 with the wave-uniform fast path: parity + sweep against round 5's rules (both orders of the variants)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_walk_gpu.py tests/test_parity_configs_gpu.py tests/test_trace_gpu.py -m gpu -q 2>&1 | grep -v "^W2026" | tail -5 > gpurun_out/r06m_tests.txt; cat gpurun_out/r06m_tests.txt
timeout 900 python profiles/r06e_sweep.py 7 C2,C4,C5 r6: r5rules:cert_ends=0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06m_sweep.txt; cut -c1-260 gpurun_out/r06m_sweep.txt
timeout 900 python profiles/r06e_sweep.py 7 C2,C4 r5rules:cert_ends=0 r6: 2>&1 | grep -v amdgpu.ids > gpurun_out/r06m_sweep_reversed.txt; cut -c1-260 gpurun_out/r06m_sweep_reversed.txt
