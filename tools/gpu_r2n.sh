#!/bin/bash
# round 2: MFMA dependency lab, layer-norm backward (rows per workgroup), C3 bench
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 120 ./tools/mfma_lab > $OUT/r2n_mfma_lab.txt 2>&1 || { tail -20 $OUT/r2n_mfma_lab.txt; exit 1; }
cat $OUT/r2n_mfma_lab.txt
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "layer_norm" > $OUT/r2n_k.log 2>&1 || { tail -50 $OUT/r2n_k.log; exit 1; }
tail -2 $OUT/r2n_k.log
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q > $OUT/r2n_m.log 2>&1 || { tail -50 $OUT/r2n_m.log; exit 1; }
tail -2 $OUT/r2n_m.log
timeout -k 10 300 python3 bench.py --model PISTRec --items 1000000 --seq-len 100 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r2n_c3.json 2> $OUT/r2n_c3.err || { tail -30 $OUT/r2n_c3.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2n_c3.json')); print('C3:', d['ms_per_step'], 'ms/step')"
