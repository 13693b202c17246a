# round 6: rule D (inverted clusters of up to three hits): parity tests, remaining literal reasons, in-process sweep vs round 5's rules,
# the aimed fuzzer on all three families again
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_walk_gpu.py tests/test_parity_configs_gpu.py tests/test_trace_gpu.py -m gpu -q 2>&1 | grep -v "^W2026" | tail -8 > gpurun_out/r06l_tests.txt; cat gpurun_out/r06l_tests.txt
TETRANERF_HIP_LIB=tetra-nerf_amd/variants/libtetranerf_hip_diag.so timeout 600 python profiles/r06b_literal_reasons.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06l_literal_reasons.txt; cat gpurun_out/r06l_literal_reasons.txt | cut -c1-200
timeout 900 python profiles/r06e_sweep.py 7 C2,C4,C5 r6: r5rules:cert_ends=0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06l_sweep.txt; cut -c1-260 gpurun_out/r06l_sweep.txt
sed -i 's/r06j_fuzz/r06l_fuzz/g' profiles/r06j_call.sh; bash profiles/r06j_call.sh
