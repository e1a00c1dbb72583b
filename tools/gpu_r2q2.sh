#!/bin/bash
# round 2: split scoring kernels (two-role forward, three-role backward) -- parity, stamps, timing
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "score32" > $OUT/r2q_k.log 2>&1 || { tail -60 $OUT/r2q_k.log; exit 1; }
tail -2 $OUT/r2q_k.log
timeout -k 10 200 ./tools/score_lab 10000003 > $OUT/r2q_lab.txt 2>&1 || { tail $OUT/r2q_lab.txt; exit 1; }
cat $OUT/r2q_lab.txt
rm -f $OUT/r2q_time.txt
for V in 1000003 10000003; do
timeout -k 10 200 python3 tools/score32_time.py $V 2>&1 | grep -v amdgpu.ids >> $OUT/r2q_time.txt || exit 1
done
cat $OUT/r2q_time.txt
