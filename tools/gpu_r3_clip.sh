#!/bin/bash
# round 3: the clip scale formed inside the optimizer launch (no ticket launch) -- tests, then the step with and without
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "clip or adam" > gpurun_out/cl_kernels.log 2>&1 || { tail -30 gpurun_out/cl_kernels.log; exit 1; }
tail -2 gpurun_out/cl_kernels.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -q -x > gpurun_out/cl_model.log 2>&1 || { tail -30 gpurun_out/cl_model.log; exit 1; }
tail -2 gpurun_out/cl_model.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-scale-legs > gpurun_out/cl_bench.json 2> gpurun_out/cl_bench.err || { tail -20 gpurun_out/cl_bench.err; exit 1; }
MTAM_CLIP_IN_ADAM=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-scale-legs > gpurun_out/cl_bench_off.json 2> gpurun_out/cl_bench_off.err || { tail -20 gpurun_out/cl_bench_off.err; exit 1; }
python - <<'PY'
import json
for f in ("cl_bench", "cl_bench_off"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["loss_first"], d["loss_last"])
PY
