set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python profiles/r06e_sweep.py 5 C2,C4,C5 default: f2k:fill_blocks=2048 f8k:fill_blocks=8192 f32k:fill_blocks=32768 f160k:fill_blocks=1000000 f8k_s2k:fill_blocks=8192,spec_blocks=2048 f8k_s8k:fill_blocks=8192,spec_blocks=8192 f32k_s32k:fill_blocks=32768,spec_blocks=32768 f8k_nospec:fill_blocks=8192,spec_fill=0 f32k_nospec:fill_blocks=32768,spec_fill=0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06t_fill_blocks_sweep.txt; cut -c1-150 gpurun_out/r06t_fill_blocks_sweep.txt
