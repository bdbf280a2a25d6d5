#!/bin/bash
# A/B of builds of the library on one GPU box (through gpurun, from the repo root):
#   bash tools/ab_bench.sh rounds contexts variants/a.so variants/b.so [...]
# alternates `bench.py` between them (TINYRENDERER_HIP_LIB) and prints frames/s and stage times.
R=$1; CTX=$2; shift 2
for i in $(seq 1 "$R"); do
  for lib in "$@"; do
    TR_BENCH_NO_VERIFY=${NO_VERIFY:-0} TINYRENDERER_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --contexts "$CTX" --steps 6 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/ab.json 2>/tmp/ab.err || { cat /tmp/ab.err; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print('$lib', round(d['value']), {k: round(v,1) for k,v in d['stages_us'].items() if v})"
  done
done
