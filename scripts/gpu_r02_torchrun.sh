#!/bin/bash
set -u
OUT=gpurun_out/r02tr; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 3 --fanout --no-cpu-baseline > $OUT/torchrun_fanout.json 2> $OUT/torchrun_fanout.err; echo "torchrun fanout rc=$?"
grep '^{' $OUT/torchrun_fanout.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('value %.2f G' % (j['value']/1e9), 'n_gpus', j['n_gpus'], 'ms/step %.4f' % j['ms_per_step'], j['config']['parallelism'][:60])"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/torchrun_plain.json 2> $OUT/torchrun_plain.err; echo "torchrun plain rc=$?"
grep '^{' $OUT/torchrun_plain.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('value %.2f G' % (j['value']/1e9), 'n_gpus', j['n_gpus'], 'ms/step %.4f' % j['ms_per_step'])"
