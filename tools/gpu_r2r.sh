#!/bin/bash
# round 2: SQ counters of the split scoring kernels at 10 M rows
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/r2r_pmc1 $OUT/r2r_pmc2
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/r2r_pmc1 -o run -- python3 tools/score32_pmc.py 10000003 > $OUT/r2r_1.log 2>&1 || { tail -20 $OUT/r2r_1.log; exit 1; }
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $OUT/r2r_pmc2 -o run -- python3 tools/score32_pmc.py 10000003 > $OUT/r2r_2.log 2>&1 || { tail -20 $OUT/r2r_2.log; exit 1; }
ls $OUT/r2r_pmc1/* $OUT/r2r_pmc2/* | head
