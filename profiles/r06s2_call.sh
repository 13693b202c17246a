set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
export TETRANERF_HIP_LIB=tetra-nerf_amd/variants/libtetranerf_hip_fillx.so
for i in 1 2 3; do timeout 600 python profiles/r06s_fill_order.py 5 0,0x40000000,0x80000000,0x100000000,0x200000000,0x40000008,0x80000008,0x100000008,0x200000008 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06s_fill_order3.txt 2>&1
cat gpurun_out/r06s_fill_order3.txt | cut -c1-260
