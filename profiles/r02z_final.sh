# final state of round 2b: GPU suite, smoke, bench line (run through gpurun from the repository root)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v "^W2026" | grep -E "passed|failed|error|^[0-9.]+s " | tail -10 > gpurun_out/r02z_gpu_tests.txt; cat gpurun_out/r02z_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r02z_smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02z_bench.json 2> gpurun_out/r02z_bench.err; tail -c 300 gpurun_out/r02z_bench.json
