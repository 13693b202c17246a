set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: beside:full_first=2 beside_nt:full_first=4 before:full_first=3"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06aj_full_first_sweep.txt; cut -c1-250 gpurun_out/r06aj_full_first_sweep.txt
