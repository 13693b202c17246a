# round 4, call ZZ: the final code once more: GPU suite, smoke, bench line (the kernels are those of call Z; since then: cross-check
# injection option, asynchronous hit-fraction read-back, debug cache check, tests)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" > gpurun_out/r04zz_gpu_tests.txt; tail -8 gpurun_out/r04zz_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04zz_bench.json 2> gpurun_out/r04zz_bench.err; tail -c 300 gpurun_out/r04zz_bench.json
