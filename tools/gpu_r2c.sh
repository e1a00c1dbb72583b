#!/bin/bash
# round 2, call c: the two-barrier GRU kernels -- parity tests, headline bench, kernel trace
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "gru or family or golden or forward or adam or trains" > $OUT/r2c_tests.log 2>&1 || { tail -40 $OUT/r2c_tests.log; exit 1; }
tail -2 $OUT/r2c_tests.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/r2c_bench.json 2> $OUT/r2c_bench.err || { tail -30 $OUT/r2c_bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2c_bench.json')); print('headline', d['ms_per_step'], 'ms/step', d['value'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2c_prof -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/r2c_prof.json 2> $OUT/r2c_prof.err || { tail -30 $OUT/r2c_prof.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/r2c_prof/*kernel_stats.csv | head -1) "bench.py --steps 100 --warmup 20 --no-cpu-baseline" > $OUT/r2c_prof.md; head -26 $OUT/r2c_prof.md
