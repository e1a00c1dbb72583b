#!/bin/bash
# round 2, call b: streaming top-K, large-catalog logits-free fp32 scoring, full-size tests, C3 / C4 benches
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "topk or score32" > $OUT/r2b_kernels.log 2>&1 || { tail -40 $OUT/r2b_kernels.log; exit 1; }
tail -2 $OUT/r2b_kernels.log
timeout -k 10 900 python3 -m pytest tests/test_full_size_gpu.py -m gpu -x -q --durations=6 > $OUT/r2b_full.log 2>&1 || { tail -60 $OUT/r2b_full.log; exit 1; }
tail -10 $OUT/r2b_full.log
timeout -k 10 300 python3 bench.py --model PISTRec --items 1000000 --seq-len 100 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/r2b_c3.json 2> $OUT/r2b_c3.err || { tail -30 $OUT/r2b_c3.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2b_c3.json')); print('C3', d['ms_per_step'], 'ms/step')"
timeout -k 10 300 python3 bench.py --items 10000000 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r2b_c4.json 2> $OUT/r2b_c4.err || { tail -30 $OUT/r2b_c4.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2b_c4.json')); print('C4 shape 1 GPU fp32', d['ms_per_step'], 'ms/step')"
MTAM_SCORE32_MAX_WGS=256 timeout -k 10 300 python3 bench.py --items 10000000 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r2b_c4_256.json 2> $OUT/r2b_c4_256.err || { tail -30 $OUT/r2b_c4_256.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2b_c4_256.json')); print('C4 256 wgs', d['ms_per_step'], 'ms/step')"
MTAM_SCORE32_MAX_WGS=2048 timeout -k 10 300 python3 bench.py --items 10000000 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r2b_c4_2048.json 2> $OUT/r2b_c4_2048.err || { tail -30 $OUT/r2b_c4_2048.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2b_c4_2048.json')); print('C4 2048 wgs', d['ms_per_step'], 'ms/step')"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2b_prof_c4 -o run -- python3 bench.py --items 10000000 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r2b_prof_c4.json 2> $OUT/r2b_prof_c4.err || { tail -30 $OUT/r2b_prof_c4.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/r2b_prof_c4/*kernel_stats.csv | head -1) "bench.py --items 10000000 --steps 10 --warmup 3" > $OUT/r2b_prof_c4.md; head -16 $OUT/r2b_prof_c4.md
