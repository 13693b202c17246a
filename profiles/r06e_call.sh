# round 6: rules A-C of the order test (end-of-chain patterns certified) + the pipelined writer: parity (the cross-check of the
# two classifications is the literal kernel itself: identical rows with cert_ends 0 / 1), literal reasons that remain, and the
# in-process interleaved sweep of the options
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_walk_gpu.py tests/test_parity_configs_gpu.py tests/test_trace_gpu.py -m gpu -q -x 2>&1 | grep -v "^W2026" | tail -15 > gpurun_out/r06e_tests.txt; cat gpurun_out/r06e_tests.txt
TETRANERF_HIP_LIB=tetra-nerf_amd/variants/libtetranerf_hip_diag.so timeout 600 python profiles/r06b_literal_reasons.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06e_literal_reasons.txt; cat gpurun_out/r06e_literal_reasons.txt
timeout 1500 python profiles/r06e_sweep.py 7 C2,C4,C5 r5rules_classic:cert_ends=0,writer_pipe=0 classic:writer_pipe=0 pipe: r5rules_pipe:cert_ends=0 beside:fill_beside_writer=1 2>&1 | grep -v amdgpu.ids > gpurun_out/r06e_sweep.txt; cat gpurun_out/r06e_sweep.txt
