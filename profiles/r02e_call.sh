cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for tree in "" "_old/"; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $R/${tree}profiles/c5_once.py 45000 2 frame > /tmp/log.txt 2>&1
  echo "=== tree '${tree}' c4 frame"; python $R/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -14
done
cd $R
python _old/profiles/r02_walk_sweep.py c4 2>&1 | grep -v "^W2026" | head -4
python profiles/r02_walk_sweep.py c4 2>&1 | grep -v "^W2026" | head -4
