# round 4, call E: both writer tables (WalkCold below 500k tets, WalkTet above): tests, A/B vs the round-3 writer, C5 per-kernel times
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_walk_gpu.py tests/test_build_gpu.py -q -x -m gpu > gpurun_out/r04e_tests.txt 2>&1; tail -4 gpurun_out/r04e_tests.txt
timeout 600 python profiles/r04d_lib_ab.py <baseline build>.so tetra-nerf_amd/libtetranerf_hip.so 3 > gpurun_out/r04e_lib_ab.txt 2>&1; cat gpurun_out/r04e_lib_ab.txt
cd /tmp
rm -rf /tmp/p5; AMD_SERIALIZE_KERNEL=3 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o t -- python $GRAFT_REPO_ROOT/profiles/c5_once.py 150000 3 1048576 > $GRAFT_REPO_ROOT/gpurun_out/r04e_c5_once.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p5 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04e_c5_serial_kernel_stats.txt
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/r04e_c5_once.txt; grep -E "k_trace_walk|k_write_segments|k_postprocess_log|k_trace_general|k_fill_range|k_verify" gpurun_out/r04e_c5_serial_kernel_stats.txt | cut -c1-60,112-160
