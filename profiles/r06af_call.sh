set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
V="default: bulk64:spec_fill=2,spec_k0=64 bulk128:spec_fill=2,spec_k0=128 bulk192:spec_fill=2,spec_k0=192 bulk256:spec_fill=2,spec_k0=256"
for c in C2 C4; do timeout 900 python profiles/r06t_alloc_sweep.py 8 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06af_bulk_rows_sweep.txt; cut -c1-250 gpurun_out/r06af_bulk_rows_sweep.txt
