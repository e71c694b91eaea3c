#!/bin/bash
# round 6: self-fold stress -- blocking ticks on a self-fold world against a k_gen_finalize shadow, every Checksum(u128) of every tick
out=gpurun_out/r06y; mkdir -p $out
timeout 900 python scripts/ff_stress.py 300000 400000 --sync > $out/selffold_stress_300k.json 2> $out/stress.err; echo "rc=$?"; cat $out/selffold_stress_300k.json | cut -c1-500
timeout 900 python scripts/ff_stress.py 1000000 150000 --sync > $out/selffold_stress_1m.json 2>> $out/stress.err; echo "rc=$?"; cat $out/selffold_stress_1m.json | cut -c1-500
