#!/bin/bash
# quick: kernel + model tests, headline bench, trace
set -o pipefail
TAG=${1:-r2q}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q > $OUT/${TAG}_t.log 2>&1 || { tail -50 $OUT/${TAG}_t.log; exit 1; }
tail -2 $OUT/${TAG}_t.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -30 $OUT/${TAG}_bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/${TAG}_bench.json')); print('headline', d['ms_per_step'], 'ms/step', d['value'], 'host-inclusive', d['host_inclusive']['value'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_prof.json 2> $OUT/${TAG}_prof.err || { tail -30 $OUT/${TAG}_prof.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/${TAG}_prof/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline" > $OUT/${TAG}_prof.md; head -16 $OUT/${TAG}_prof.md
