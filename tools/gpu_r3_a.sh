#!/bin/bash
# round 3, session A: GPU tests, headline bench (all legs), kernel trace of the same command, PMC passes of the
# roofline_at_scale cases (FETCH_SIZE and WRITE_SIZE in separate runs, nothing else traced)
set -o pipefail
TAG=${1:-r3a}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1 || { tail -40 $OUT/${TAG}_pytest.log; exit 1; }
tail -3 $OUT/${TAG}_pytest.log
fi
timeout -k 10 500 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -30 $OUT/${TAG}_bench.err; exit 1; }
python3 - <<PY
import json
d = json.load(open('$OUT/${TAG}_bench.json'))
print('headline', round(d['ms_per_step'], 4), 'ms', round(d['value']), 'seq/s; host-inclusive', round(d['host_inclusive']['value']), 'cpu', round(d['cpu_baseline']['value']))
print('gather B=128 frac', round(d['roofline']['frac'], 3), 'scatter', round(d['roofline_scatter_add']['frac'], 3))
for r in d.get('roofline_at_scale', []):
    print(' ', r['kernel'], r['sequences_per_launch'], r['item_rows'], r['id_dist'], 'us', round(r['us_per_launch'], 1), 'frac', round(r['frac'], 3), 'sets', r['buffer_sets'])
print('gru', d.get('gru_serial_model'))
print('fused fwd', d.get('roofline_fused_forward', {}).get('us_per_launch'))
PY
if [ "${SKIP_PROF:-0}" != "1" ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- python3 bench.py --no-cpu-baseline --no-scale-legs > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err || { tail -30 $OUT/${TAG}_prof.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/${TAG}_prof/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-scale-legs" > $OUT/${TAG}_kernel_stats.md
head -24 $OUT/${TAG}_kernel_stats.md | cut -c1-160
fi
if [ "${SKIP_PMC:-0}" != "1" ]; then
CASES=""
for c in "512 3709 zipf" "2048 3709 zipf" "512 1000003 zipf" "2048 1000003 zipf" "512 1000003 uniform" "2048 1000003 uniform" "128 3709 zipf"; do
  set -- $c; name=B$1_V$2_$3
  for ctr in FETCH_SIZE WRITE_SIZE; do
    kind=$( [ $ctr = FETCH_SIZE ] && echo fetch || echo write )
    timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d $OUT/${TAG}_pmc/pmc_${kind}_$name -o run -- python3 tools/emb_roofline.py pmc $1 $2 4 $3 > $OUT/${TAG}_pmc_${kind}_$name.log 2>&1 || { tail -20 $OUT/${TAG}_pmc_${kind}_$name.log; exit 1; }
  done
  CASES="$CASES,$name"
  echo "pmc $name done"
done
python3 tools/summarize_prof.py pmc_scale $OUT/${TAG}_pmc ${CASES#,} > $OUT/${TAG}_pmc_emb_scale.json
python3 -c "
import json
d = json.load(open('$OUT/${TAG}_pmc_emb_scale.json'))
for k, v in d.items():
    print(k, {n: round(x['traffic_bytes'] / 1e6, 2) for n, x in v.items()})
"
fi
