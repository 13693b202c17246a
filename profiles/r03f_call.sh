# round 3, call F: debugging (poisoned scratch), hole fuzz with rule 8, GPU suite with poisoned outputs
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TETRANERF_HIP_POISON=1 timeout 300 python profiles/r03f_debug.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03f_debug.txt; cat gpurun_out/r03f_debug.txt
timeout 600 python profiles/r03_hole_fuzz.py 50 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03f_hole_fuzz.txt; grep -c MISMATCH gpurun_out/r03f_hole_fuzz.txt; tail -2 gpurun_out/r03f_hole_fuzz.txt
TETRANERF_HIP_POISON=1 timeout 1800 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^W2026" > gpurun_out/r03f_gpu_tests_poison.txt; tail -22 gpurun_out/r03f_gpu_tests_poison.txt
