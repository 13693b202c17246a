set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: bulk32:spec_fill=2,spec_k0=32 bulk64:spec_fill=2,spec_k0=64 bulk96:spec_fill=2,spec_k0=96 bulk128:spec_fill=2,spec_k0=128 bulk192:spec_fill=2,spec_k0=192 bulk_auto:spec_fill=2"
for c in C2 C4; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06ad_bulk_sweep.txt; cut -c1-250 gpurun_out/r06ad_bulk_sweep.txt
V="default: bulk128:spec_fill=2,spec_k0=128 bulk192:spec_fill=2,spec_k0=192 bulk256:spec_fill=2,spec_k0=256 bulk320:spec_fill=2,spec_k0=320 bulk_auto:spec_fill=2"
timeout 900 python profiles/r06t_alloc_sweep.py 4 C5 $V 2>&1 | grep -v amdgpu.ids > gpurun_out/r06ad_bulk_sweep_c5.txt; cut -c1-250 gpurun_out/r06ad_bulk_sweep_c5.txt
