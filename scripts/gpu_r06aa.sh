#!/bin/bash
# round 6: the ALU ceiling with the kernels' own diffuse, and config 5 on the same box
out=gpurun_out/r06final; mkdir -p $out
./scripts/ubench_alu > $out/ubench_alu.txt 2>&1; tail -14 $out/ubench_alu.txt
timeout 600 python bench.py --config 5 --steps 20 --warmup 3 > $out/bench_config5_1gpu_samebox.json 2>> $out/bench.err; python -c "
import json; j=json.loads(open('$out/bench_config5_1gpu_samebox.json').read()); print(j['value']/1e9, j['roofline_alu'])"
