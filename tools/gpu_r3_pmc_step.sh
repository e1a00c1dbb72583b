#!/bin/bash
# PMC traffic (FETCH_SIZE, WRITE_SIZE in separate passes, nothing else traced) of the training step's kernels
set -o pipefail
TAG=${1:-r3pmc}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  kind=$( [ $ctr = FETCH_SIZE ] && echo fetch || echo write )
  MTAM_HIP_GRAPH=0 timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d $OUT/${TAG}_$kind -o run -- python3 bench.py --no-cpu-baseline --no-scale-legs --steps 20 --warmup 5 > $OUT/${TAG}_$kind.json 2> $OUT/${TAG}_$kind.err || { tail -20 $OUT/${TAG}_$kind.err; exit 1; }
done
python3 tools/summarize_prof.py pmc_step $(ls $OUT/${TAG}_fetch/*counter_collection.csv | head -1) $(ls $OUT/${TAG}_write/*counter_collection.csv | head -1) "seq_chain,tagru,weight_grads,emb_scatter,ta_attn,adam_kernel,bwd_tr3,lse_kernel,sqnorm" > $OUT/${TAG}_pmc_step.json
python3 -c "
import json
d = json.load(open('$OUT/${TAG}_pmc_step.json'))
for k, v in d.items():
    print(k[:60].ljust(60), v['launches'], 'read MB', round(v['read_bytes_corrected'] / 1e6, 2), 'write MB', round(v['write_bytes'] / 1e6, 2))
"
