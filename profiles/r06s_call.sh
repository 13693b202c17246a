set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( while true; do date +%s.%N; rocm-smi --showclocks --showpower --showmemuse 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|power\|GPU Memory Allocated" ; sleep 1; done ) > gpurun_out/r06s_clocks.txt 2>&1 &
SMI=$!
for i in 1 2 3 4 5 6; do date +%s.%N; timeout 300 python profiles/r06s_placement.py 4 0 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06s_placement.txt 2>&1
for i in 1 2; do date +%s.%N; timeout 300 python profiles/r06s_placement.py 5 1537 2>&1 | grep -v amdgpu.ids; done >> gpurun_out/r06s_placement.txt 2>&1
kill $SMI
cat gpurun_out/r06s_placement.txt | cut -c1-260
rocm-smi --showmemvendor --showvbios 2>/dev/null | tail -5
cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size 2>/dev/null
