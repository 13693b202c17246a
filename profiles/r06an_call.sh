# SQ counters of the walk kernel on the C2 frame: where do its wave cycles go (active VALU issue, parked on s_waitcnt, issue stalls)?
set -x
cd /tmp; export TMPDIR=/tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out
B3="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-configs --no-calibration --no-c6"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/q$i; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/q$i -o c -- $B3 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/q$i -name "*.db" | head -1) | grep -E "k_trace_walk|k_hull_entry|k_write_segments|k_fill_rows_fine" >> $GRAFT_REPO_ROOT/gpurun_out/r06an_walk_sq_counters.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r06an_walk_sq_counters.txt | cut -c1-40,82-180
