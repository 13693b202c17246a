# round 3, call G: who is right when walk and BVH disagree (brute-force oracle), sampler / train / bf16x3 tests
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python profiles/r03_hole_classify.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03g_hole_classify.txt; tail -30 gpurun_out/r03g_hole_classify.txt
timeout 900 python -m pytest tests/test_samplers_gpu.py tests/test_train_gpu.py tests/test_sharding_gpu.py tests/test_render_gpu.py -m gpu -q 2>&1 | grep -v "^W2026" > gpurun_out/r03g_tests.txt; tail -15 gpurun_out/r03g_tests.txt
