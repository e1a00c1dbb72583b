#!/bin/bash
# round 3: two ranks sharing the one GPU of the box over gloo -- a rehearsal of `python bench.py --gpus 2` with the
# real kernels and world_size 2 (launcher, rank code, every exchange); timings mean nothing
set -o pipefail
TAG=${1:-r3dp2}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export MTAM_BENCH_SHARE_GPU=1
run() { name=$1; shift; timeout -k 10 400 "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || { tail -40 $OUT/${TAG}_$name.err; exit 1; }; python3 -c "import json; d=json.load(open('$OUT/${TAG}_$name.json')); print('$name', 'ranks', d['ranks_seen'], round(d['ms_per_step'],3), 'ms/step', (d['exchange'] or 'single GPU')[:44], 'loss', round(d['loss_first'],4), '->', round(d['loss_last'],4))"; }
run flat_ml1m python3 bench.py --gpus 2 --steps 40 --warmup 10 --no-scale-legs
run flat python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-scale-legs --items 200000 --dp-exchange flat
run sharded python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-scale-legs --items 200000 --dp-exchange sharded
run sharded_scoring python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-scale-legs --items 200000 --dp-exchange sharded-scoring
run sharded_table python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-scale-legs --items 200000 --dp-exchange sharded-table
python3 - <<PY
import json
L = {k: json.load(open('$OUT/${TAG}_%s.json' % k)) for k in ('flat', 'sharded', 'sharded_scoring')}
ref = L['flat']
for k, d in L.items():
    for f in ('loss_first', 'loss_last'):
        rel = abs(d[f] - ref[f]) / abs(ref[f])
        assert rel < 2e-5, (k, f, d[f], ref[f])
    assert d['ranks_seen'] == 2 and len(d['ms_per_step_by_rank']) == 2
print('the four exchanges agree: loss', ref['loss_first'], '->', ref['loss_last'], '(2 ranks, 200,000 items, 13 steps)')
PY
# ... and four ranks (row ranges of a quarter each; five processes on the card)
run flat4 python3 bench.py --gpus 4 --steps 6 --warmup 2 --no-scale-legs --items 100003 --dp-exchange flat
run sharded4 python3 bench.py --gpus 4 --steps 6 --warmup 2 --no-scale-legs --items 100003 --dp-exchange sharded
run sharded_scoring4 python3 bench.py --gpus 4 --steps 6 --warmup 2 --no-scale-legs --items 100003 --dp-exchange sharded-scoring
run sharded_table4 python3 bench.py --gpus 4 --steps 6 --warmup 2 --no-scale-legs --items 100003 --dp-exchange sharded-table
python3 - <<PY
import json
L = {k: json.load(open('$OUT/${TAG}_%s.json' % k)) for k in ('flat4', 'sharded4', 'sharded_scoring4')}
ref = L['flat4']
for k, d in L.items():
    for f in ('loss_first', 'loss_last'):
        assert abs(d[f] - ref[f]) / abs(ref[f]) < 2e-5, (k, f, d[f], ref[f])
    assert d['ranks_seen'] == 4
print('the four exchanges agree: loss', ref['loss_first'], '->', ref['loss_last'], '(4 ranks, 100,003 items, 8 steps)')
PY
