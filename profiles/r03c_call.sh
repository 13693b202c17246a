# round 3, call C: GPU suite (pruned code + dynamic group hand-out in the writer + plugin / large-M / C4-size backward tests),
# writer shapes again, PMC pass of the writer
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python profiles/r03b_writer.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03c_writer.txt; cat gpurun_out/r03c_writer.txt
timeout 1800 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "^W2026" | tail -60 > gpurun_out/r03c_gpu_tests.txt; tail -30 gpurun_out/r03c_gpu_tests.txt
cd /tmp
for v in "c2 x=0" "c4 x=0"; do
  set -- $v; tag=$1
  if [ "$tag" = "c4" ]; then mesh="45000 2"; else mesh="15000 0"; fi
  rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $mesh frame > /tmp/log.txt 2>&1
  ( grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -8 ) > $GRAFT_REPO_ROOT/gpurun_out/r03c_${tag}_timeline.txt
done
rm -rf /tmp/pc; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU -d /tmp/pc -o c -- python $GRAFT_REPO_ROOT/profiles/c5_once.py 15000 0 frame > /tmp/logc.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/pc -name "*.db" | head -1) | grep -E "^#|^kernel|k_trace_walk|k_write_segments|k_fill_range|k_postprocess_log" > $GRAFT_REPO_ROOT/gpurun_out/r03c_pmc_sq.txt
cat $GRAFT_REPO_ROOT/gpurun_out/r03c_*_timeline.txt; cut -c1-50,82-145 $GRAFT_REPO_ROOT/gpurun_out/r03c_pmc_sq.txt
