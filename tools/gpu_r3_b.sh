#!/bin/bash
# round 3, quick loop: selected GPU tests, then the headline bench without the slow legs
set -o pipefail
TAG=${1:-r3b}; shift
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q "$@" > $OUT/${TAG}_pytest.log 2>&1 || { tail -60 $OUT/${TAG}_pytest.log; exit 1; }
tail -3 $OUT/${TAG}_pytest.log
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-scale-legs > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -30 $OUT/${TAG}_bench.err; exit 1; }
python3 - <<PY
import json
d = json.load(open('$OUT/${TAG}_bench.json'))
print('headline', round(d['ms_per_step'], 4), 'ms', round(d['value']), 'seq/s; host-inclusive', round(d['host_inclusive']['value']), round(d['host_inclusive']['ms_per_step'], 4))
print('fused fwd us', d.get('roofline_fused_forward', {}).get('us_per_launch'), 'gru', d.get('gru_serial_model', {}).get('us_fwd'), d.get('gru_serial_model', {}).get('us_bwd'))
PY
