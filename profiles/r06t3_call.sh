set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: f32k_nospec:fill_blocks=32768,spec_fill=0 f16k_nospec:fill_blocks=16384,spec_fill=0 f64k_nospec:fill_blocks=65536,spec_fill=0 f160k_nospec:fill_blocks=1000000,spec_fill=0 f32k_nospec_w0:fill_blocks=32768,spec_fill=0,walk_lds_kb=0 fs32k:fill_blocks=32768,spec_blocks=32768 fs160k:fill_blocks=1000000,spec_blocks=1000000"
for c in C2 C4 C5; do timeout 900 python profiles/r06t_alloc_sweep.py 4 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06t_alloc_sweep2.txt; cut -c1-250 gpurun_out/r06t_alloc_sweep2.txt
