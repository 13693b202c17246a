# round 6: the pipelined segment writer: parity test (vs the classic writer + the full-size oracle comparisons), then the
# interleaved A/B: base (round 5 kernels) | w3 classic writer | w3 pipelined | w3 pipelined + tail fill beside the writer
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V=tetra-nerf_amd/variants
timeout 900 python -m pytest tests/test_walk_gpu.py tests/test_parity_configs_gpu.py tests/test_trace_gpu.py -m gpu -q -x 2>&1 | grep -v "^W2026" | tail -15 > gpurun_out/r06d_tests.txt; cat gpurun_out/r06d_tests.txt
timeout 1500 python profiles/r06c_lib_ab.py 3 C2,C4,C5 base=$V/libtetranerf_hip_base.so classic=$V/libtetranerf_hip_w3.so:writer_pipe=0 pipe=$V/libtetranerf_hip_w3.so:writer_pipe=1 beside=$V/libtetranerf_hip_w3.so:writer_pipe=1:fill_beside_writer=1 2>&1 | grep -v amdgpu.ids > gpurun_out/r06d_lib_ab.txt; cat gpurun_out/r06d_lib_ab.txt
