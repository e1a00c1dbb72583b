#!/bin/bash
# round 2: full GPU suite, then the large configurations with the three-role scoring backward; C3 and C4 kernel traces
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/r2s_tests.log 2>&1 || { tail -60 $OUT/r2s_tests.log; exit 1; }
tail -2 $OUT/r2s_tests.log
run() { name=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > $OUT/r2s_$name.json 2> $OUT/r2s_$name.err || { tail -30 $OUT/r2s_$name.err; exit 1; }; python3 -c "import json; d=json.load(open('$OUT/r2s_$name.json')); print('$name', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'seq/s')"; }
run c3 --model PISTRec --items 1000000 --seq-len 100 --steps 30 --warmup 5
run c4 --items 10000000 --steps 20 --warmup 5
run c5_f32 --items 50000000 --seq-len 200 --steps 6 --warmup 2
prof() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2s_prof_$name -o run -- python3 bench.py --no-cpu-baseline "$@" > $OUT/r2s_prof_$name.json 2> $OUT/r2s_prof_$name.err || { tail -30 $OUT/r2s_prof_$name.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/r2s_prof_$name/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline $*" > $OUT/r2s_prof_$name.md; head -12 $OUT/r2s_prof_$name.md; }
prof c3 --model PISTRec --items 1000000 --seq-len 100 --steps 20 --warmup 5
prof c4 --items 10000000 --steps 10 --warmup 3
