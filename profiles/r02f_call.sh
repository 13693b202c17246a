cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_render_gpu.py -x -q -m gpu 2>&1 | grep -v "^W2026" | tail -15 | tee gpurun_out/r02f_render_tests.txt
python profiles/r02f_sched_sweep.py c2 c4 c5 2>&1 | grep -v "^W2026" | tee gpurun_out/r02f_sched_sweep.txt
