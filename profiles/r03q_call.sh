# round 3, call Q: what k_mlp_backward / k_mlp_forward / k_dw_gemm wait for (SQ counters, three separate passes)
set -x
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pq$i -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/pq$i -name "*.db" | head -1) | grep -E "^#|^kernel|k_mlp_backward|k_mlp_forward<true, false|k_dw_gemm<4, false" > $GRAFT_REPO_ROOT/gpurun_out/r03q_pmc_$i.txt
done
cd $GRAFT_REPO_ROOT; cat gpurun_out/r03q_pmc_*.txt | cut -c1-30,100-200
