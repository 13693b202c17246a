set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( for i in 1 2; do TN_FILL_ROWS_FINE=2 python profiles/r06s_torch_fill.py 6 | sed 's/tn_fill_rows/units tn_fill_rows/'; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06u_units_fill.txt; cut -c1-100 gpurun_out/r06u_units_fill.txt
V="default: fine_nospec:fill_blocks=-1,spec_fill=0 units_nospec:fill_blocks=-2,spec_fill=0"
for c in C2 C4; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06u_alloc_sweep2.txt; cut -c1-250 gpurun_out/r06u_alloc_sweep2.txt
