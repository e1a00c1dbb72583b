#!/bin/bash
# round 2, call f: LDS-staged 16-wave GRU kernels -- parity, stamps, bench
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "gru or family or golden or forward or adam or trains" > $OUT/r2f_tests.log 2>&1 || { tail -60 $OUT/r2f_tests.log; exit 1; }
tail -2 $OUT/r2f_tests.log
timeout -k 5 120 ./tools/gru_lab > $OUT/r2f_gru_lab.log 2>&1; cat $OUT/r2f_gru_lab.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/r2f_bench.json 2> $OUT/r2f_bench.err || { tail -30 $OUT/r2f_bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2f_bench.json')); print('headline', d['ms_per_step'], 'ms/step', d['value'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2f_prof -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/r2f_prof.json 2> $OUT/r2f_prof.err || { tail -30 $OUT/r2f_prof.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/r2f_prof/*kernel_stats.csv | head -1) "bench.py --steps 100 --warmup 20 --no-cpu-baseline" > $OUT/r2f_prof.md; head -14 $OUT/r2f_prof.md
