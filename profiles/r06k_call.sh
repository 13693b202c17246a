# round 6: full GPU suite (bf16x3 one-launch render, stride 256, rules A-C) + the aimed certification fuzzer on old and new families
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" | tail -30 > gpurun_out/r06k_tests.txt; tail -12 gpurun_out/r06k_tests.txt
bash profiles/r06j_call.sh
