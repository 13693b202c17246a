# round 6, final build: the aimed certification fuzzer at four times the usual volume (flat hull search, rules A-D, fine tail fill)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python profiles/r03_hole_fuzz.py 400 0 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06ar_fuzz_default_noverify.txt; tail -1 gpurun_out/r06ar_fuzz_default_noverify.txt | cut -c1-600
timeout 1500 python profiles/r03_hole_fuzz.py 200 1 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06ar_fuzz_default_verify.txt; tail -1 gpurun_out/r06ar_fuzz_default_verify.txt | cut -c1-600
TETRANERF_FUZZ_FAMILY=planar timeout 1500 python profiles/r03_hole_fuzz.py 400 0 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r06ar_fuzz_planar_noverify.txt; tail -1 gpurun_out/r06ar_fuzz_planar_noverify.txt | cut -c1-600
TETRANERF_FUZZ_FAMILY=planar timeout 1500 python profiles/r03_hole_fuzz.py 200 1 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r06ar_fuzz_planar_verify.txt; tail -1 gpurun_out/r06ar_fuzz_planar_verify.txt | cut -c1-600
TETRANERF_FUZZ_FAMILY=big timeout 2400 python profiles/r03_hole_fuzz.py 60 0 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06ar_fuzz_big_noverify.txt; tail -1 gpurun_out/r06ar_fuzz_big_noverify.txt | cut -c1-600
