#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bench; mkdir -p $O
( time timeout 900 python bench.py --steps 10 --warmup 3 ) > $O/n1.json 2> $O/n1.err
echo "rc=$?" >> $O/n1.err
export HDK_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --rows 256000000 > $O/n2_c2.json 2> $O/n2_c2.err
echo "rc=$?" >> $O/n2_c2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --rows 128000000 --config c5 --no-cpu-baseline > $O/n2_c5.json 2> $O/n2_c5.err
echo "rc=$?" >> $O/n2_c5.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --rows 128000000 --config q3 --scaling weak > $O/n2_q3.json 2> $O/n2_q3.err
echo "rc=$?" >> $O/n2_q3.err
