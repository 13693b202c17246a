# round 6, closing call (after the fine tail fill and the hull-entry kernel): GPU suite, smoke, the bench line, rocprofv3 kernel stats + PMC traffic of the bench command, kernel stats
# of a training iteration and of one rendered frame, one-launch vs kernel-chain render A/B (65,536- and 4,096-ray chunks)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "^W2026" > gpurun_out/r06zz_gpu_tests.txt; tail -14 gpurun_out/r06zz_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/r06zz_bench_full.json > gpurun_out/r06zz_bench.json 2> gpurun_out/r06zz_bench.err; cat gpurun_out/r06zz_bench.json
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-render --no-configs --no-calibration"
rm -rf /tmp/p1; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o s -- $B > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p1 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r06zz_trace_kernel_stats.txt
B3="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-configs --no-calibration"
rm -rf /tmp/p2; timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p2 -o w -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p2 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r06zz_trace_pmc_write.txt
rm -rf /tmp/p3; timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o f -- $B3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py pmc $(find /tmp/p3 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r06zz_trace_pmc_fetch.txt
rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03ae_train_tn.py > /tmp/train.log 2>&1
( grep "ms per iteration" /tmp/train.log; python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) ) > $GRAFT_REPO_ROOT/gpurun_out/r06zz_train_tn_kernel_stats.txt
rm -rf /tmp/p5; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o r -- python $GRAFT_REPO_ROOT/profiles/r05_render_ab.py 1 > /tmp/render.log 2>&1
( grep -v "^W2026\|^E2026\|amdgpu.ids" /tmp/render.log; python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p5 -name "*.db" | head -1) ) > $GRAFT_REPO_ROOT/gpurun_out/r06zz_render_kernel_stats.txt
cd $GRAFT_REPO_ROOT
python profiles/r05_render_ab.py 3 65536 2>&1 | grep -v amdgpu.ids > gpurun_out/r06zz_render_ab.txt
python profiles/r05_render_ab.py 3 4096 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06zz_render_ab.txt
python profiles/r05_render_ab.py 3 65536 bf16x3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06zz_render_ab.txt
cat gpurun_out/r06zz_render_ab.txt
head -12 gpurun_out/r06zz_trace_kernel_stats.txt | cut -c1-60,112-160
grep -E "k_trace_walk|k_write_seg|k_fill|k_postprocess_log|k_trace_general|k_verify" gpurun_out/r06zz_trace_pmc_write.txt gpurun_out/r06zz_trace_pmc_fetch.txt | cut -c1-90,110-220
head -14 gpurun_out/r06zz_train_tn_kernel_stats.txt | cut -c1-70,112-160
