set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "45000 2 frame c4frame" "150000 3 1048576 c5rays" "45000 2 4096 c4batch" "150000 3 frame c5frame"; do
  set -- $cfg
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $R/profiles/c5_once.py $1 $2 $3 > $R/gpurun_out/r02b_$4.log 2>&1
  python $R/profiles/rocprof_summary.py stats $(find /tmp/pp -name "*.db" | head -1) > $R/gpurun_out/r02b_$4_kernel_stats.txt
  head -12 $R/gpurun_out/r02b_$4_kernel_stats.txt | cut -c1-60,112-160; tail -1 $R/gpurun_out/r02b_$4.log
done
