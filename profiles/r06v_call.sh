set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_configs_gpu.py tests/test_walk_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06v_tests.txt; cat gpurun_out/r06v_tests.txt
V="default: round5_schedule:fill_blocks=512,spec_fill=1 w1k:writer_blocks=1024 w2k:writer_blocks=2048 w4k:writer_blocks=4096 wall:writer_blocks=100000"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 4 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06v_writer_grid.txt; cut -c1-250 gpurun_out/r06v_writer_grid.txt
