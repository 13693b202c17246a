set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: xcd_span:fill_blocks=-3"
for c in C2 C4 C2; do timeout 900 python profiles/r06t_alloc_sweep.py 8 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06bb_xcd_span_sweep.txt; cut -c1-250 gpurun_out/r06bb_xcd_span_sweep.txt
