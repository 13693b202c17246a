cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_build_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^W2026" | tail -25 | tee gpurun_out/r02b_build_tests.txt
cd /tmp
for cfg in "15000 0 frame c2frame_v1 seg_variant=1" "15000 0 frame c2frame_v0 seg_variant=0" "45000 2 frame c4frame_v1 seg_variant=1" "45000 2 frame c4frame_v0 seg_variant=0" "150000 3 1048576 c5rays_v1 seg_variant=1"; do
  set -- $cfg
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $R/profiles/c5_once.py $1 $2 $3 $5 > $R/gpurun_out/r02c_$4.log 2>&1
  python $R/profiles/rocprof_summary.py stats $(find /tmp/pp -name "*.db" | head -1) > $R/gpurun_out/r02c_$4_kernel_stats.txt
  echo $4; head -8 $R/gpurun_out/r02c_$4_kernel_stats.txt | cut -c1-50,112-160; grep -v "^W2026" $R/gpurun_out/r02c_$4.log | tail -1
done
