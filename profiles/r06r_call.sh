set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python profiles/r06e_sweep.py 7 C2,C4,C5 default: nospec:spec_fill=0 walk8:walk_lds_kb=0 spec512:spec_blocks=256 2>&1 | grep -v amdgpu.ids > gpurun_out/r06r_spec_sweep.txt; cut -c1-150 gpurun_out/r06r_spec_sweep.txt
TETRANERF_HIP_LIB=tetra-nerf_amd/variants/libtetranerf_hip_thin.so timeout 900 python profiles/r06e_sweep.py 5 C2,C4,C5 thin_default: thin_nospec:spec_fill=0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06r_thin_sweep.txt; cut -c1-250 gpurun_out/r06r_thin_sweep.txt
