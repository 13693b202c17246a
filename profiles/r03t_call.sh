# round 3, call T: full GPU suite on the new training kernels + render bench (one launch per pass vs separate kernels)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" > gpurun_out/r03t_gpu_tests.txt; tail -9 gpurun_out/r03t_gpu_tests.txt
timeout 600 python profiles/r02j_render_bench.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03t_render_bench.txt; cat gpurun_out/r03t_render_bench.txt
