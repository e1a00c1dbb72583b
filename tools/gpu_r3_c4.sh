#!/bin/bash
# round 3: large-catalog configurations on one GPU, incl. the data-parallel rehearsals (one-rank RCCL group)
set -o pipefail
TAG=${1:-r3c4}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 500 "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || { tail -30 $OUT/${TAG}_$name.err; exit 1; }; python3 -c "import json; d=json.load(open('$OUT/${TAG}_$name.json')); print('$name', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'seq/s', d.get('exchange'))"; }
B="python3 bench.py --no-cpu-baseline --no-scale-legs"
run c3 $B --model PISTRec --items 1000000 --seq-len 100 --steps 30 --warmup 5
run c4 $B --items 10000000 --steps 20 --warmup 5
export MTAM_BENCH_FORCE_DP=1
run c4_dp1_sharded $B --items 10000000 --steps 20 --warmup 5 --dp-exchange sharded
run c4_dp1_sharded_scoring $B --items 10000000 --steps 20 --warmup 5 --dp-exchange sharded-scoring
run c4_dp1_sharded_table $B --items 10000000 --steps 20 --warmup 5 --dp-exchange sharded-table
run c2_dp1_flat $B --steps 200 --warmup 20 --dp-exchange flat
run c2_dp1_sharded_scoring $B --steps 200 --warmup 20 --dp-exchange sharded-scoring
