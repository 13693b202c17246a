set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: f4k:fill_blocks=4096 f8k:fill_blocks=8192 f32k:fill_blocks=32768 f160k:fill_blocks=1000000 fs8k:fill_blocks=8192,spec_blocks=8192 fs32k:fill_blocks=32768,spec_blocks=32768 f32k_nospec:fill_blocks=32768,spec_fill=0"
for c in C2 C4 C2; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06t_alloc_sweep.txt; cut -c1-200 gpurun_out/r06t_alloc_sweep.txt
