#!/bin/bash
for w in 512 58 39 29; do
  MTAM_SCORE32_MAX_WGS=$w python3 bench.py --no-cpu-baseline --no-scale-legs --steps 600 --warmup 50 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('max_wgs', $w, round(d['ms_per_step'],4))"
done
