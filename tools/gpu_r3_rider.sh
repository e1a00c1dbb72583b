#!/bin/bash
# round 3: the clip's partial pass riding in the scatter-add launch -- tests, then the step with and without
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "scatter or clip or adam" > gpurun_out/rd_kernels.log 2>&1 || { tail -30 gpurun_out/rd_kernels.log; exit 1; }
tail -2 gpurun_out/rd_kernels.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -q -x > gpurun_out/rd_model.log 2>&1 || { tail -30 gpurun_out/rd_model.log; exit 1; }
tail -2 gpurun_out/rd_model.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-scale-legs > gpurun_out/rd_bench.json 2> gpurun_out/rd_bench.err || { tail -20 gpurun_out/rd_bench.err; exit 1; }
MTAM_NORM_RIDER=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-scale-legs > gpurun_out/rd_bench_off.json 2> gpurun_out/rd_bench_off.err || { tail -20 gpurun_out/rd_bench_off.err; exit 1; }
python - <<'PY'
import json
for f in ("rd_bench", "rd_bench_off"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["loss_first"], d["loss_last"])
PY
