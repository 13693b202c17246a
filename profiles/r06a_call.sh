# round 6, first call: GPU suite, smoke, the new bench line (short stdout line + full JSON, C6 leg)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | grep -v "^W2026" > gpurun_out/r06a_gpu_tests.txt; tail -14 gpurun_out/r06a_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1200 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/r06a_bench_full.json > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err ) 2>&1 | tail -4
wc -c gpurun_out/r06a_bench.json; cat gpurun_out/r06a_bench.json; tail -5 gpurun_out/r06a_bench.err | cut -c1-400
