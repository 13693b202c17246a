set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( for i in 1 2; do python profiles/r06s_torch_fill.py 5; TN_FILL_ROWS_FINE=1 python profiles/r06s_torch_fill.py 5 | sed 's/tn_fill_rows/fine tn_fill_rows/'; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06u_fine_fill.txt; cut -c1-100 gpurun_out/r06u_fine_fill.txt
V="default: f160k_nospec:fill_blocks=1000000,spec_fill=0 fine_nospec:fill_blocks=-1,spec_fill=0 fine_spec512:fill_blocks=-1 fine_specfine:fill_blocks=-1,spec_blocks=-1"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06u_alloc_sweep.txt; cut -c1-250 gpurun_out/r06u_alloc_sweep.txt
