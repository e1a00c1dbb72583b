#!/bin/bash
# round 2: fused gather forward -- kernel test, model tests, headline bench + trace
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "seq_chain or scatter or gather or score32" > $OUT/r2m_k.log 2>&1 || { tail -50 $OUT/r2m_k.log; exit 1; }
tail -2 $OUT/r2m_k.log
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "not c5" > $OUT/r2m_m.log 2>&1 || { tail -50 $OUT/r2m_m.log; exit 1; }
tail -2 $OUT/r2m_m.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/r2m_bench.json 2> $OUT/r2m_bench.err || { tail -30 $OUT/r2m_bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2m_bench.json')); print('fused gather:', d['ms_per_step'], 'ms/step', d['value'], 'host-inclusive', d['host_inclusive']['value'])"
MTAM_FUSED_GATHER=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/r2m_bench_nofuse.json 2> $OUT/r2m_bench_nofuse.err || { tail -30 $OUT/r2m_bench_nofuse.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r2m_bench_nofuse.json')); print('two kernels :', d['ms_per_step'], 'ms/step', d['value'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2m_prof -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/r2m_prof.json 2> $OUT/r2m_prof.err || { tail -30 $OUT/r2m_prof.err; exit 1; }
python3 tools/summarize_prof.py stats $(ls $OUT/r2m_prof/*kernel_stats.csv | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline" > $OUT/r2m_prof.md; head -14 $OUT/r2m_prof.md
