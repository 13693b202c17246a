set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: beside_after:fill_beside=1 beside_before:fill_beside=2 beside_after_512:fill_beside=1,fill_blocks=512 beside_after_256:fill_beside=1,fill_blocks=256"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 4 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06w_fill_beside.txt; cut -c1-250 gpurun_out/r06w_fill_beside.txt
