cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v "^W2026" | tail -14 > gpurun_out/r02g_gpu_tests.txt; cat gpurun_out/r02g_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r02g_smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02g_bench.json 2> gpurun_out/r02g_bench.err; tail -c 600 gpurun_out/r02g_bench.json; tail -3 gpurun_out/r02g_bench.err
