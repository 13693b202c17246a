# round 3, call A: schedule experiments + timelines + PMC diagnostics of the walk / segment writer
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python profiles/r03a_sched.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03a_sched.txt; cat gpurun_out/r03a_sched.txt
cd /tmp
for v in "c2base" "c2fuse fuse_fill=1 spec_fill=0" "c4fuse fuse_fill=1 spec_fill=0"; do
  set -- $v; tag=$1; shift
  if [ "${tag#c4}" != "$tag" ]; then mesh="45000 2"; else mesh="15000 0"; fi
  rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $mesh frame "$@" > /tmp/log.txt 2>&1
  ( grep -v "^W2026\|^E2026" /tmp/log.txt | tail -1; python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -8 ) > $GRAFT_REPO_ROOT/gpurun_out/r03a_${tag}_timeline.txt
done
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rm -rf /tmp/pc; timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pc -o c -- python $GRAFT_REPO_ROOT/profiles/c5_once.py 15000 0 frame > /tmp/logc.txt 2>&1
  python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/pc -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r03a_pmc_$i.txt 2>&1 || tail -5 /tmp/logc.txt > $GRAFT_REPO_ROOT/gpurun_out/r03a_pmc_$i.txt
done
cd $GRAFT_REPO_ROOT; cat gpurun_out/r03a_*_timeline.txt; cat gpurun_out/r03a_pmc_*.txt | cut -c1-60,82-150
