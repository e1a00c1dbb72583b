#!/bin/bash
# round 3: other shapes through the step (mixed precision at L = 200, two decoder blocks) -- do they still run, what do they take
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; shift; python3 bench.py --no-cpu-baseline --no-scale-legs "$@" 2> $OUT/regress_$name.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), 'ms/step, loss', round(d['loss_first'],4), '->', round(d['loss_last'],4))" || { tail -20 $OUT/regress_$name.err; exit 1; }; }
run c5like_bf16 --items 5000000 --seq-len 200 --score-dtype bf16 --steps 5 --warmup 2
run c5like_f32 --items 5000000 --seq-len 200 --steps 5 --warmup 2
run nb2_h2 --blocks 2 --heads 2 --steps 50 --warmup 10
run pistrec_small --model PISTRec --seq-len 100 --steps 30 --warmup 5
