#!/bin/bash
# round 2: full GPU suite with the split-bf16 GEMM path; headline, C3, C4 benches
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/r2w_tests.log 2>&1 || { tail -60 $OUT/r2w_tests.log; exit 1; }
tail -2 $OUT/r2w_tests.log
run() { name=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > $OUT/r2w_$name.json 2> $OUT/r2w_$name.err || { tail -30 $OUT/r2w_$name.err; exit 1; }; python3 -c "import json; d=json.load(open('$OUT/r2w_$name.json')); print('$name', round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'seq/s')"; }
run headline
MTAM_GEMM_SPLIT=0 run headline_fp32gemm
run c3 --model PISTRec --items 1000000 --seq-len 100 --steps 30 --warmup 5
MTAM_GEMM_SPLIT=0 run c3_fp32gemm --model PISTRec --items 1000000 --seq-len 100 --steps 30 --warmup 5
run c4 --items 10000000 --steps 20 --warmup 5
