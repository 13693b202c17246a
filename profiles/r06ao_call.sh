set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: units2:writer_units=2 units2_b2k:writer_units=2,writer_blocks=2048"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06ao_writer_units_sweep.txt; cut -c1-250 gpurun_out/r06ao_writer_units_sweep.txt
timeout 600 python profiles/r06e_sweep.py 3 C2,C4,C5 default: units2:writer_units=2 2>&1 | grep -v amdgpu.ids > gpurun_out/r06ao_writer_units_breakdown.txt; cut -c1-330 gpurun_out/r06ao_writer_units_breakdown.txt
