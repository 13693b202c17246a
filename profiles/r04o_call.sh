# round 4, call O: C5 (1M tets, 2^20 rays): memory counters of the walk and the segment writer with the per-tet writer table
# (same recipe as profiles/r03ab_call.sh, whose r03ab_c5_pmc_2.txt is the round-3 reference: writer 275 M fetches beyond the L2)
set -x
cd /tmp; export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pq$i -o t -- python $GRAFT_REPO_ROOT/profiles/c5_once.py 150000 3 1048576 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/pq$i -name "*.db" | head -1) | grep -E "^#|^kernel|k_trace_walk|k_write_segments|k_postprocess_log|k_fill_range|k_verify" > $GRAFT_REPO_ROOT/gpurun_out/r04o_c5_pmc_$i.txt
done
cd $GRAFT_REPO_ROOT; cat gpurun_out/r04o_c5_pmc_*.txt | cut -c1-34,62-150
