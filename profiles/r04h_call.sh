# round 4, call H: mid-round check of the whole GPU suite, smoke and the bench line (new fields: mesh hashes, cross-check counters)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "^W2026" > gpurun_out/r04h_gpu_tests.txt; tail -14 gpurun_out/r04h_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04h_bench.json 2> gpurun_out/r04h_bench.err; tail -c 600 gpurun_out/r04h_bench.json; tail -3 gpurun_out/r04h_bench.err
