# does an allocator setting change the placement spread?  C2 frame per fresh allocation under three environments
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default:"
( echo "== plain"; timeout 600 python profiles/r06t_alloc_sweep.py 10 C2 $V 2>&1 | grep -v amdgpu.ids
  echo "== PYTORCH_HIP_ALLOC_CONF=expandable_segments:True"; PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True timeout 600 python profiles/r06t_alloc_sweep.py 10 C2 $V 2>&1 | grep -v amdgpu.ids | tail -12
  echo "== HSA_DISABLE_FRAGMENT_ALLOCATOR=1"; HSA_DISABLE_FRAGMENT_ALLOCATOR=1 timeout 600 python profiles/r06t_alloc_sweep.py 10 C2 $V 2>&1 | grep -v amdgpu.ids | tail -12
  echo "== PYTORCH_NO_HIP_MEMORY_CACHING=1"; PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python profiles/r06t_alloc_sweep.py 6 C2 $V 2>&1 | grep -v amdgpu.ids | tail -8
) > gpurun_out/r06aw_allocator_env.txt 2>&1
cut -c1-120 gpurun_out/r06aw_allocator_env.txt
