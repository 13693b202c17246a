set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( for g in 0 1 3 7; do TN_FILL_ROWS_STREAM=$g python profiles/r06s_torch_fill.py 4 | sed "s/tn_fill_rows/stream$g tn_fill_rows/"; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06ah_stream_fill.txt; cut -c1-60 gpurun_out/r06ah_stream_fill.txt
V="default: stream1k:fill_blocks=-3 stream2k:fill_blocks=-4 stream4k:fill_blocks=-6 stream8k:fill_blocks=-10"
for c in C2 C4; do timeout 900 python profiles/r06t_alloc_sweep.py 6 $c $V 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06ah_stream_sweep.txt; cut -c1-250 gpurun_out/r06ah_stream_sweep.txt
