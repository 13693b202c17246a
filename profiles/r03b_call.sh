# round 3, call B: GPU suite on the pruned code + segment-writer shapes + timelines
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python profiles/r03b_writer.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03b_writer.txt; cat gpurun_out/r03b_writer.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "^W2026" | tail -40 > gpurun_out/r03b_gpu_tests.txt; tail -25 gpurun_out/r03b_gpu_tests.txt
cd /tmp
for v in "c2u4 seg_unroll=4" "c2u2 seg_unroll=2" "c4u2 seg_unroll=2"; do
  set -- $v; tag=$1; shift
  if [ "${tag#c4}" != "$tag" ]; then mesh="45000 2"; else mesh="15000 0"; fi
  rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $mesh frame "$@" > /tmp/log.txt 2>&1
  ( grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -8 ) > $GRAFT_REPO_ROOT/gpurun_out/r03b_${tag}_timeline.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r03b_*_timeline.txt
