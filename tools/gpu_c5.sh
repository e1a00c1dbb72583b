#!/bin/bash
# bf16-scoring session: parity at small and full sizes, then step times at C4 / C5 shapes in both modes.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "bf16 or score16 or adam" > gpurun_out/c5_tests_small.log 2>&1 || { tail -30 gpurun_out/c5_tests_small.log; exit 1; }
tail -2 gpurun_out/c5_tests_small.log
timeout -k 10 900 python -m pytest tests/test_full_size_gpu.py -x -q -k "c5_bf16" > gpurun_out/c5_tests_full.log 2>&1 || { tail -40 gpurun_out/c5_tests_full.log; exit 1; }
tail -2 gpurun_out/c5_tests_full.log
for dt in bf16 f32; do
  timeout -k 10 600 python bench.py --items 10000000 --score-dtype $dt --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c4_$dt.json 2> gpurun_out/bench_c4_$dt.log || { tail -20 gpurun_out/bench_c4_$dt.log; exit 1; }
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_c4_$dt.json')); print('C4 shape', '$dt', d['ms_per_step'], 'ms/step', d['loss_first'], d['loss_last'])"
done
for dt in bf16 f32; do
  timeout -k 10 900 python bench.py --items 50000000 --seq-len 200 --score-dtype $dt --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c5_$dt.json 2> gpurun_out/bench_c5_$dt.log || { tail -20 gpurun_out/bench_c5_$dt.log; exit 1; }
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_c5_$dt.json')); print('C5 shape', '$dt', d['ms_per_step'], 'ms/step', d['loss_first'], d['loss_last'])"
done
