cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_configs_gpu.py tests/test_walk_gpu.py tests/test_trace_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r02b_parity.txt
python profiles/r02b_writer_ablate.py c2 c4 c5 2>&1 | grep -v "^W2026" | tee gpurun_out/r02b_writer_variants.txt
