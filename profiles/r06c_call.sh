# round 6: why rays are literal (diag build) + interleaved A/B of: base (round 5 kernels), w1 (walk: unconditional log store),
# w2 (w1 + writer: all LDS reads, then all stores)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V=tetra-nerf_amd/variants
TETRANERF_HIP_LIB=$V/libtetranerf_hip_diag.so timeout 600 python profiles/r06b_literal_reasons.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06b_literal_reasons.txt; cat gpurun_out/r06b_literal_reasons.txt
timeout 1500 python profiles/r06c_lib_ab.py 3 C2,C4,C5 base=$V/libtetranerf_hip_base.so w1=$V/libtetranerf_hip_w1.so w2=$V/libtetranerf_hip_w2.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r06c_lib_ab.txt; cat gpurun_out/r06c_lib_ab.txt
