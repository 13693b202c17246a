# round 6: (1) the train tests with the re-based float64 comparison, (2) blind cross-check stride 1024 vs 256 vs 64 in-process,
# (3) C5 memory counters of the walk / writer / literal pairing on the round-6 kernels (recipe of profiles/r04o_call.sh)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_walk_gpu.py -m gpu -q -s 2>&1 | grep -v "^W2026" | grep "ReLU decisions\|passed\|failed\|Error\|assert" | cut -c1-400 > gpurun_out/r06i_tests.txt; cat gpurun_out/r06i_tests.txt
timeout 900 python profiles/r06e_sweep.py 7 C2,C4,C5 stride1024: stride256:verify_stride=256 stride64:verify_stride=64 r5rules:cert_ends=0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06i_stride_sweep.txt; cat gpurun_out/r06i_stride_sweep.txt | cut -c1-200
cd /tmp
i=0
for C in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pq$i -o t -- python $GRAFT_REPO_ROOT/profiles/c5_once.py 150000 3 1048576 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/pq$i -name "*.db" | head -1) | grep -E "^#|^kernel|k_trace_walk|k_write_segments|k_postprocess_log|k_fill_range|k_verify|k_trace_general" > $GRAFT_REPO_ROOT/gpurun_out/r06i_c5_pmc_$i.txt
done
cd $GRAFT_REPO_ROOT; cat gpurun_out/r06i_c5_pmc_*.txt | cut -c1-34,62-150
