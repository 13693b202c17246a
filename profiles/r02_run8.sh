set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_gather_gpu.py tests/test_render_gpu.py tests/test_reference_suite_gpu.py tests/test_trace_gpu.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r02_run8_tests.txt
cat gpurun_out/r02_run8_tests.txt
timeout 300 python profiles/r02_train_bench.py > gpurun_out/r02_train_bench.txt 2>&1
cat gpurun_out/r02_train_bench.txt
timeout 300 python profiles/r02_gather_match_bench.py > gpurun_out/r02_gather_match.txt 2>&1
cat gpurun_out/r02_gather_match.txt
