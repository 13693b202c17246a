# round 3, call S: C5 (1M tets, 2^20 rays) with the kernels of a call SERIALISED (AMD_SERIALIZE_KERNEL=3): what each one
# costs alone, against the concurrent schedule
set -x
cd /tmp; export TMPDIR=/tmp
for mode in concurrent serial; do
  if [ $mode = serial ]; then export AMD_SERIALIZE_KERNEL=3; fi
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py 150000 3 1048576 > /tmp/log.txt 2>&1
  ( echo "# $mode"; grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -5 ) > $GRAFT_REPO_ROOT/gpurun_out/r03s_c5_$mode.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r03s_c5_*.txt
