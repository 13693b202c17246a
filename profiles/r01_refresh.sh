set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r01_gpu_tests.txt; cat gpurun_out/r01_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r01_bench.json 2> gpurun_out/r01_bench.err; tail -c 600 gpurun_out/r01_bench.json
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render"
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r01_trace_kernel_stats.txt
rm -rf /tmp/p2; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p2 -o w -- $B > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r01_trace_pmc_write.txt
rm -rf /tmp/p3; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o f -- $B > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p3 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r01_trace_pmc_fetch.txt
rm -rf /tmp/p4; rocprofv3 --kernel-trace --stats -d /tmp/p4 -o m -- python $GRAFT_REPO_ROOT/profiles/mlp_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r01_mlp_bench.log 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r01_mlp_kernel_stats.txt
rm -rf /tmp/p5; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 -d /tmp/p5 -o q -- python $GRAFT_REPO_ROOT/profiles/mlp_bench.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p5 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r01_mlp_pmc_mfma.txt
head -12 $GRAFT_REPO_ROOT/gpurun_out/r01_trace_kernel_stats.txt; cat $GRAFT_REPO_ROOT/gpurun_out/r01_trace_pmc_write.txt $GRAFT_REPO_ROOT/gpurun_out/r01_trace_pmc_fetch.txt | cut -c1-200; tail -12 $GRAFT_REPO_ROOT/gpurun_out/r01_mlp_pmc_mfma.txt | cut -c1-200
