#!/bin/bash
# round 2, final numbers: headline bench (all legs), large configurations, C3 / C4 kernel traces
set -o pipefail
TAG=${1:-r2f}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -30 $OUT/${TAG}_bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/${TAG}_bench.json')); print('headline', d['ms_per_step'], d['value'], 'host-inclusive', d['host_inclusive']['value'], 'cpu', d['cpu_baseline']['value'])"
run() { name=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || { tail -30 $OUT/${TAG}_$name.err; exit 1; }; python3 -c "import json; d=json.load(open('$OUT/${TAG}_$name.json')); print('$name', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'seq/s')"; }
run c3 --model PISTRec --items 1000000 --seq-len 100 --steps 30 --warmup 5
run c4 --items 10000000 --steps 20 --warmup 5
run c5_f32 --items 50000000 --seq-len 200 --steps 6 --warmup 2
run c5_bf16 --items 50000000 --seq-len 200 --score-dtype bf16 --steps 6 --warmup 2
prof() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$name -o run -- python3 bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_prof_$name.json 2> $OUT/${TAG}_prof_$name.err || { tail -30 $OUT/${TAG}_prof_$name.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/${TAG}_prof_$name/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline $*" > $OUT/${TAG}_prof_$name.md; head -12 $OUT/${TAG}_prof_$name.md | cut -c1-150; }
prof c3 --model PISTRec --items 1000000 --seq-len 100 --steps 20 --warmup 5
prof c4 --items 10000000 --steps 10 --warmup 3
