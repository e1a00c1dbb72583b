#!/bin/bash
# One gpurun call: GPU parity tests, headline bench, rocprof kernel trace, PMC traffic passes, embedding sweep.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh TAG
set -o pipefail
TAG=${1:-s}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q --durations=8 > $OUT/gpu_tests_$TAG.log 2>&1 || { tail -30 $OUT/gpu_tests_$TAG.log; exit 1; }
tail -14 $OUT/gpu_tests_$TAG.log
timeout -k 10 300 python3 bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || { tail -30 $OUT/bench_$TAG.err; exit 1; }
cat $OUT/bench_$TAG.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/prof_$TAG.json 2> $OUT/prof_$TAG.err || { tail -30 $OUT/prof_$TAG.err; exit 1; }
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o run -- python3 tools/emb_roofline.py pmc 128 3709 10 > $OUT/pmc_fetch_$TAG.log 2>&1 || { tail -30 $OUT/pmc_fetch_$TAG.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o run -- python3 tools/emb_roofline.py pmc 128 3709 10 > $OUT/pmc_write_$TAG.log 2>&1 || { tail -30 $OUT/pmc_write_$TAG.log; exit 1; }
timeout -k 10 400 python3 tools/emb_roofline.py sweep > $OUT/emb_sweep_$TAG.jsonl 2> $OUT/emb_sweep_$TAG.err || { tail -30 $OUT/emb_sweep_$TAG.err; exit 1; }
cat $OUT/emb_sweep_$TAG.jsonl
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke_$TAG.log 2>&1 || { tail -30 $OUT/smoke_$TAG.log; exit 1; }
tail -1 $OUT/smoke_$TAG.log
timeout -k 10 300 python3 bench.py --model PISTRec --items 1000000 --seq-len 100 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_$TAG.json 2> $OUT/bench_c3_$TAG.err || { tail -30 $OUT/bench_c3_$TAG.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/bench_c3_$TAG.json')); print('C3', d['ms_per_step'], 'ms/step', d['value'], 'seq/s')"
