#!/bin/bash
# Validates the N > 1 bench code path on a 1-GPU box: fan-out GPU test, then bench.py --fanout both
# directly and under torch.distributed.run with one rank (the driver's launch line with N = 1).
TAG=${1:-fanout}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_zfanout.py -m gpu -x -q 2>&1 | tail -12 | tee $OUT/pytest.log
timeout 300 python bench.py --fanout --steps 200 --warmup 16 > $OUT/bench_fanout.json 2> $OUT/err.log; echo "rc=$?"; cat $OUT/bench_fanout.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --fanout --steps 200 --warmup 16 > $OUT/bench_fanout_torchrun.json 2>> $OUT/err.log; echo "rc=$?"; cat $OUT/bench_fanout_torchrun.json
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2>> $OUT/err.log; cat $OUT/bench.json
tail -5 $OUT/err.log
