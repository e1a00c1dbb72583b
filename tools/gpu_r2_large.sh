#!/bin/bash
# round 2: the large-catalog configurations (C3, C4 shape, C5 fp32 / bf16) with the split-bf16 fp32 scoring; C4 kernel trace
set -o pipefail
TAG=${1:-r2l}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || { tail -30 $OUT/${TAG}_$name.err; exit 1; }; python3 -c "import json; d=json.load(open('$OUT/${TAG}_$name.json')); print('$name', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'seq/s; host-inclusive', round(d.get('host_inclusive',{}).get('ms_per_step',0),3))"; }
run c3 --model PISTRec --items 1000000 --seq-len 100 --steps 30 --warmup 5
run c4 --items 10000000 --steps 20 --warmup 5
MTAM_SCORE32_SPLIT_MIN_ROWS=0 run c4_native --items 10000000 --steps 10 --warmup 3
run c5_f32 --items 50000000 --seq-len 200 --steps 6 --warmup 2
run c5_bf16 --items 50000000 --seq-len 200 --score-dtype bf16 --steps 6 --warmup 2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c4 -o run -- python3 bench.py --items 10000000 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_prof_c4.json 2> $OUT/${TAG}_prof_c4.err || { tail -30 $OUT/${TAG}_prof_c4.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/${TAG}_prof_c4/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --items 10000000 --steps 10 --warmup 3 --no-cpu-baseline" > $OUT/${TAG}_prof_c4.md; head -14 $OUT/${TAG}_prof_c4.md
