cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_render_gpu.py tests/test_sharding_gpu.py tests/test_walk_gpu.py -q -m gpu 2>&1 | grep -v "^W2026" | tail -5
python profiles/r02j_render_bench.py 2>&1 | grep -v "^W2026" | tee gpurun_out/r02j_render_bench.txt
