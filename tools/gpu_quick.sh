#!/bin/bash
# Short gpurun call: GPU parity tests + headline bench (no CPU baseline) + optional extra command.
set -o pipefail
TAG=${1:-q}; shift
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests_$TAG.log 2>&1 || { tail -40 $OUT/gpu_tests_$TAG.log; exit 1; }
tail -2 $OUT/gpu_tests_$TAG.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || { tail -30 $OUT/bench_$TAG.err; exit 1; }
cat $OUT/bench_$TAG.json
for cmd in "$@"; do
  echo "== $cmd"
  timeout -k 10 500 bash -c "$cmd" || exit 1
done
