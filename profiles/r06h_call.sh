set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python profiles/r06h_grad_bisect.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06h_grad_bisect.txt; cat gpurun_out/r06h_grad_bisect.txt
