cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python profiles/r02d_pipe_sweep.py c2 c4 c5 2>&1 | grep -v "^W2026" | tee gpurun_out/r02d_pipe_sweep.txt
python -m pytest tests/test_build_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^W2026" | tail -4 | tee gpurun_out/r02d_build_tests.txt
