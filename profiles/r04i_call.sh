# round 4, call I: per-dispatch timelines of the C2 / C4 frames with the round-4 schedule (2 + 6 blocks per CU, late verify)
set -x
cd /tmp; export TMPDIR=/tmp
for cfg in "15000 0 frame c2frame" "45000 2 frame c4frame"; do
  set -- $cfg
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $1 $2 $3 > /tmp/log.txt 2>&1
  ( grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -10 ) > $GRAFT_REPO_ROOT/gpurun_out/r04i_$4_timeline.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r04i_*_timeline.txt
