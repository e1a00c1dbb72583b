#!/bin/bash
# round 3: the one-launch small-catalog scoring kernel -- its test under a short timeout first, then the suites it touches
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "score32" > gpurun_out/fs_kernels.log 2>&1 || { tail -30 gpurun_out/fs_kernels.log; exit 1; }
tail -3 gpurun_out/fs_kernels.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -q -x > gpurun_out/fs_model.log 2>&1 || { tail -30 gpurun_out/fs_model.log; exit 1; }
tail -3 gpurun_out/fs_model.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-scale-legs > gpurun_out/fs_bench.json 2> gpurun_out/fs_bench.err || { tail -20 gpurun_out/fs_bench.err; exit 1; }
MTAM_SCORE32_FUSED=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-scale-legs > gpurun_out/fs_bench_off.json 2> gpurun_out/fs_bench_off.err || { tail -20 gpurun_out/fs_bench_off.err; exit 1; }
python - <<'PY'
import json
for f in ("fs_bench", "fs_bench_off"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["loss_first"], d["loss_last"])
PY
