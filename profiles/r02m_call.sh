cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>gpurun_out/r02m_dist.err | tail -c 1500 | tee gpurun_out/r02m_dist_bench.json
tail -3 gpurun_out/r02m_dist.err
