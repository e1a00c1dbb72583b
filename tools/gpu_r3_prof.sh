#!/bin/bash
# kernel trace of the headline bench (no cpu baseline, no scale legs)
set -o pipefail
TAG=${1:-r3p}; shift
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- python3 bench.py --no-cpu-baseline --no-scale-legs "$@" > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err || { tail -30 $OUT/${TAG}_prof.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/${TAG}_prof/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-scale-legs $*" > $OUT/${TAG}_kernel_stats.md
head -26 $OUT/${TAG}_kernel_stats.md | cut -c1-150
python3 -c "
import json
d = json.load(open('$OUT/${TAG}_prof_bench.json'))
print('ms/step under trace', d['ms_per_step'], 'fused', d.get('roofline_fused_forward', {}).get('us_per_launch'))"
