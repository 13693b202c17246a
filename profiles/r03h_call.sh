# round 3, call H: hole fuzz on the shipped walk path (no cross-check) and with it; full GPU suite (poisoned outputs)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python profiles/r03_hole_fuzz.py 100 0 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03h_hole_fuzz_noverify.txt; grep -c MISMATCH gpurun_out/r03h_hole_fuzz_noverify.txt; tail -2 gpurun_out/r03h_hole_fuzz_noverify.txt
timeout 600 python profiles/r03_hole_fuzz.py 50 1 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03h_hole_fuzz_verify.txt; tail -1 gpurun_out/r03h_hole_fuzz_verify.txt
TETRANERF_HIP_POISON=1 timeout 1800 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" > gpurun_out/r03h_gpu_tests_poison.txt; tail -12 gpurun_out/r03h_gpu_tests_poison.txt
