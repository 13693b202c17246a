set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
export TETRANERF_HIP_LIB=tetra-nerf_amd/variants/libtetranerf_hip_fillx.so
for cfg in "0 512" "0 1000000" "1 512" "1 1000000" "2 1000000" "1 8192"; do set -- $cfg; echo "== TN_FILL_SEQ=$1 TN_FILL_ROWS_BLOCKS=$2"; TN_FILL_SEQ=$1 TN_FILL_ROWS_BLOCKS=$2 timeout 300 python profiles/r06s_placement.py 6 0 2>&1 | grep -v amdgpu.ids | cut -c60-; done > gpurun_out/r06s_fill_seq.txt 2>&1
cat gpurun_out/r06s_fill_seq.txt | cut -c1-200
