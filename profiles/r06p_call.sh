set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python profiles/r06e_sweep.py 7 C2,C4,C5 r6_p0: r6_p1:verify_place=1 r6_p2:verify_place=2 r6_nov:verify_stride=0 r5_p0:cert_ends=0 r5_p1:cert_ends=0,verify_place=1 r5_p2:cert_ends=0,verify_place=2 r5_nov:cert_ends=0,verify_stride=0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06p_place_sweep.txt; cut -c1-150 gpurun_out/r06p_place_sweep.txt
