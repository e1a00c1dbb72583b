#!/bin/bash
# round 2 full session: every GPU test, headline bench (all legs), kernel trace, PMC traffic passes, embedding sweep, smoke
set -o pipefail
TAG=${1:-r2s}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > $OUT/${TAG}_tests.log 2>&1 || { tail -40 $OUT/${TAG}_tests.log; exit 1; }
tail -12 $OUT/${TAG}_tests.log
timeout -k 10 400 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -30 $OUT/${TAG}_bench.err; exit 1; }
cat $OUT/${TAG}_bench.json; grep "bench\]" $OUT/${TAG}_bench.err | tail -8
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_prof.json 2> $OUT/${TAG}_prof.err || { tail -30 $OUT/${TAG}_prof.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/${TAG}_prof/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline" > $OUT/${TAG}_prof.md; head -30 $OUT/${TAG}_prof.md
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o run -- python3 tools/emb_roofline.py pmc 128 3709 10 > $OUT/${TAG}_pmc_fetch.log 2>&1 || { tail -30 $OUT/${TAG}_pmc_fetch.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o run -- python3 tools/emb_roofline.py pmc 128 3709 10 > $OUT/${TAG}_pmc_write.log 2>&1 || { tail -30 $OUT/${TAG}_pmc_write.log; exit 1; }
python3 tools/summarize_prof.py pmc $(ls $OUT/${TAG}_pmc_fetch/*counter_collection.csv | head -1) $(ls $OUT/${TAG}_pmc_write/*counter_collection.csv | head -1) > $OUT/${TAG}_pmc_emb.json; cat $OUT/${TAG}_pmc_emb.json
timeout -k 10 500 python3 tools/emb_roofline.py sweep > $OUT/${TAG}_emb_sweep.jsonl 2> $OUT/${TAG}_emb_sweep.err || { tail -30 $OUT/${TAG}_emb_sweep.err; exit 1; }
cat $OUT/${TAG}_emb_sweep.jsonl
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1 || { tail -30 $OUT/${TAG}_smoke.log; exit 1; }
tail -1 $OUT/${TAG}_smoke.log
MTAM_BENCH_FORCE_DP=1 timeout -k 10 300 python3 bench.py --items 10000000 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_c4_dp1.json 2> $OUT/${TAG}_c4_dp1.err || { tail -30 $OUT/${TAG}_c4_dp1.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/${TAG}_c4_dp1.json')); print('C4 shape, 1-rank sharded exchange rehearsal:', d['ms_per_step'], 'ms/step')"
