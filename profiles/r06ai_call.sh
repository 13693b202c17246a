set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( TN_FILL_ROWS_L4=1 python profiles/r06s_torch_fill.py 6 | sed "s/tn_fill_rows/linear4 tn_fill_rows/" ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06ai_linear4_fill.txt; cut -c1-60 gpurun_out/r06ai_linear4_fill.txt
V="default: linear4:fill_blocks=-3"
for c in C2 C4; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06ai_linear4_sweep.txt; cut -c1-250 gpurun_out/r06ai_linear4_sweep.txt
