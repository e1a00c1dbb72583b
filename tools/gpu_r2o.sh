#!/bin/bash
# round 2: cyclic slab walk in the split scoring kernels -- parity + timing against the blocked walk
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "score32" > $OUT/r2o_k.log 2>&1 || { tail -50 $OUT/r2o_k.log; exit 1; }
tail -2 $OUT/r2o_k.log
for V in 1000003 10000003; do
for cyc in 0 1; do
echo "== V=$V MTAM_SCORE32_CYCLIC=$cyc" >> $OUT/r2o_time.txt
MTAM_SCORE32_CYCLIC=$cyc timeout -k 10 200 python3 tools/score32_time.py $V 2>&1 | grep -v amdgpu.ids | head -3 >> $OUT/r2o_time.txt || exit 1
done; done
cat $OUT/r2o_time.txt
