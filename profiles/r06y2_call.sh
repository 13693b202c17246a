set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_walk_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06y_tests.txt; cat gpurun_out/r06y_tests.txt
TETRANERF_HIP_LIB=tetra-nerf_amd/variants/libtetranerf_hip_diag.so timeout 600 python profiles/r06y_walk_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y_walk_time_split.txt; cat gpurun_out/r06y_walk_time_split.txt
V="default: tree:hull_flat=0"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 3 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06y_hull_split_sweep.txt; cut -c1-250 gpurun_out/r06y_hull_split_sweep.txt
timeout 600 python profiles/r06x_batch_breakdown.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y_batch_breakdown.txt; cat gpurun_out/r06y_batch_breakdown.txt
