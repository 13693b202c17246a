cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python profiles/r02o_mlp_block.py 2>&1 | grep -v "^W2026" | tee gpurun_out/r02o_mlp_block.txt
timeout 600 python -m pytest tests/test_render_gpu.py tests/test_train_gpu.py -q -m gpu 2>&1 | grep -v "^W2026" | grep -E "passed|failed|rror" | tail -4
