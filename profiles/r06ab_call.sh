# round 6, after the flat LDS hull search (k_hull_entry) and the fine tail fill: the aimed certification fuzzer again
# order test, grouped-store writer): the families of rounds 3-5, the new planar-patch family, the 1M-point mesh
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python profiles/r03_hole_fuzz.py 100 0 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06ab_fuzz_default_noverify.txt; tail -1 gpurun_out/r06ab_fuzz_default_noverify.txt | cut -c1-600
timeout 900 python profiles/r03_hole_fuzz.py 50 1 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06ab_fuzz_default_verify.txt; tail -1 gpurun_out/r06ab_fuzz_default_verify.txt | cut -c1-600
TETRANERF_FUZZ_FAMILY=planar timeout 900 python profiles/r03_hole_fuzz.py 100 0 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r06ab_fuzz_planar_noverify.txt; tail -1 gpurun_out/r06ab_fuzz_planar_noverify.txt | cut -c1-600
TETRANERF_FUZZ_FAMILY=planar timeout 900 python profiles/r03_hole_fuzz.py 50 1 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r06ab_fuzz_planar_verify.txt; tail -1 gpurun_out/r06ab_fuzz_planar_verify.txt | cut -c1-600
TETRANERF_FUZZ_FAMILY=big timeout 1500 python profiles/r03_hole_fuzz.py 20 0 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06ab_fuzz_big_noverify.txt; tail -1 gpurun_out/r06ab_fuzz_big_noverify.txt | cut -c1-600
TETRANERF_FUZZ_FAMILY=big timeout 1500 python profiles/r03_hole_fuzz.py 10 1 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06ab_fuzz_big_verify.txt; tail -1 gpurun_out/r06ab_fuzz_big_verify.txt | cut -c1-600
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06ab_gpu_tests.txt; cat gpurun_out/r06ab_gpu_tests.txt
