set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_configs_gpu.py tests/test_walk_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06ac_tests.txt; cat gpurun_out/r06ac_tests.txt
( for i in 1 2; do TN_FILL_ROWS_LINEAR=1 python profiles/r06s_torch_fill.py 6 | sed 's/tn_fill_rows/linear tn_fill_rows/'; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06ac_linear_fill2.txt; cut -c1-120 gpurun_out/r06ac_linear_fill2.txt
V="default: fine:fill_blocks=-1 round5_schedule:fill_blocks=512,spec_fill=1"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06ac_linear_sweep2.txt; cut -c1-250 gpurun_out/r06ac_linear_sweep2.txt
