#!/bin/bash
# Every GPU test file in a process of its own (order-dependence check), one after the other.
set -o pipefail
mkdir -p gpurun_out
for f in $(grep -l "mark.gpu" tests/*.py); do
  n=$(basename $f .py)
  MTAM_SKIP_C5=${MTAM_SKIP_C5:-0} timeout -k 10 500 python3 -m pytest $f -m gpu -x -q > gpurun_out/each_$n.log 2>&1 || { echo "FAILED $f"; tail -25 gpurun_out/each_$n.log; exit 1; }
  echo "$n: $(tail -1 gpurun_out/each_$n.log)"
done
