# round 3, call D: writer static vs dynamic (same box), full GPU suite with complete failure output, hole fuzz, a bench run
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python profiles/r03d_writer_ab.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03d_writer_ab.txt; cat gpurun_out/r03d_writer_ab.txt
timeout 1800 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "^W2026" > gpurun_out/r03d_gpu_tests_full.txt; tail -25 gpurun_out/r03d_gpu_tests_full.txt
timeout 600 python profiles/r03_hole_fuzz.py 50 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03d_hole_fuzz.txt; tail -4 gpurun_out/r03d_hole_fuzz.txt
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err; tail -c 600 gpurun_out/r03d_bench.json; tail -5 gpurun_out/r03d_bench.err
