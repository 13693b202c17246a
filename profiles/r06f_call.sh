# round 6: full GPU suite on the new default (pipelined writer, rules A-C, line-aligned fill head), then the in-process sweep
# classic vs pipelined writer
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" | tail -25 > gpurun_out/r06f_tests.txt; cat gpurun_out/r06f_tests.txt
timeout 1200 python profiles/r06e_sweep.py 7 C2,C4,C5 classic:writer_pipe=0 pipe: r5rules_classic:cert_ends=0,writer_pipe=0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06f_sweep.txt; cat gpurun_out/r06f_sweep.txt
