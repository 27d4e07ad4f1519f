cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time python bench.py --steps 10) > gpurun_out/b1.json 2> gpurun_out/b1.err; tail -4 gpurun_out/b1.err; cut -c1-300 gpurun_out/b1.json
(time LVX_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2) > gpurun_out/b2.json 2> gpurun_out/b2.err; tail -6 gpurun_out/b2.err; cat gpurun_out/b2.json | cut -c1-200
