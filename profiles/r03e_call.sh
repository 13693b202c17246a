# round 3, call E: the train test alone (fresh process) and in file order, GPU suite, hole analysis, sampler tests
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -k render_train 2>&1 | grep -v "^W2026" | tail -15 > gpurun_out/r03e_train_alone.txt; tail -5 gpurun_out/r03e_train_alone.txt
timeout 1800 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "^W2026" > gpurun_out/r03e_gpu_tests_full.txt; tail -25 gpurun_out/r03e_gpu_tests_full.txt
timeout 900 python profiles/r03_hole_analyse.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03e_hole_analyse.txt; tail -40 gpurun_out/r03e_hole_analyse.txt
