# round 6: per-dispatch timelines of the C2 and C4 frames with round 5's order rules (cert_ends = 0) and round 6's -- the r6 walk is
# faster alone and the call is slower as scheduled: where?
set -x
cd /tmp; export TMPDIR=/tmp
for cfg in "15000 0 c2" "45000 2 c4"; do
  set -- $cfg
  for ends in 0 1; do
    rm -rf /tmp/tl; timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $1 $2 frame cert_ends=$ends > /tmp/tl.log 2>&1
    ( tail -1 /tmp/tl.log; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/tl -name "*.db" | head -1) 200 | tail -14 ) > $GRAFT_REPO_ROOT/gpurun_out/r06n_$3_ends${ends}_timeline.txt
    cat $GRAFT_REPO_ROOT/gpurun_out/r06n_$3_ends${ends}_timeline.txt | cut -c1-120
  done
done
