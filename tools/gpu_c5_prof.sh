#!/bin/bash
# Kernel-time table of the C5 step (bf16 scoring) and HBM traffic of the two score16 passes.
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o run -- python3 bench.py --items 50000000 --seq-len 200 --score-dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_c5.json 2> $OUT/prof_c5.err || { tail -30 $OUT/prof_c5.err; exit 1; }
cat $OUT/prof_c5.json
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_s16 -o run -- ./tools/score16_lab_base 50000003 > $OUT/pmc_fetch_s16.log 2>&1 || { tail -30 $OUT/pmc_fetch_s16.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_s16 -o run -- ./tools/score16_lab_base 50000003 > $OUT/pmc_write_s16.log 2>&1 || { tail -30 $OUT/pmc_write_s16.log; exit 1; }
tail -3 $OUT/pmc_write_s16.log
