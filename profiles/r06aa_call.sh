set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: noprio:walk_prio=0"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 3 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06aa_prio_sweep.txt; cut -c1-250 gpurun_out/r06aa_prio_sweep.txt
timeout 600 python profiles/r06e_sweep.py 3 C2,C4,C5 default: noprio:walk_prio=0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06aa_prio_breakdown.txt; cut -c1-330 gpurun_out/r06aa_prio_breakdown.txt
