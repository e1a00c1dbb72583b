#!/bin/bash
# round 2: split-bf16 operands in the generic GEMM and the grouped weight-gradient launch
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or weight_grads or grouped" > $OUT/r2u_k.log 2>&1 || { tail -60 $OUT/r2u_k.log; exit 1; }
tail -2 $OUT/r2u_k.log
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q > $OUT/r2u_m.log 2>&1 || { tail -60 $OUT/r2u_m.log; exit 1; }
tail -2 $OUT/r2u_m.log
for sp in 1 0; do
MTAM_GEMM_SPLIT=$sp timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/r2u_bench_$sp.json 2> $OUT/r2u_bench_$sp.err || { tail -30 $OUT/r2u_bench_$sp.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2u_bench_$sp.json')); print('MTAM_GEMM_SPLIT=$sp:', d['ms_per_step'], 'ms/step', d['value'])"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2u_prof -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/r2u_prof.json 2> $OUT/r2u_prof.err || { tail -30 $OUT/r2u_prof.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/r2u_prof/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline" > $OUT/r2u_prof.md; head -22 $OUT/r2u_prof.md | cut -c1-150
