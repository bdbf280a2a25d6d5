#!/bin/bash
# exact vs fused colour arithmetic of k_shade, and build variants of the fused kernel, on one box:
#   gpurun -- bash tools/ab_shade.sh rounds lib.so [lib.so ...]
R=$1; shift
for i in $(seq 1 "$R"); do
  for lib in "$@"; do
    for ex in 1 0; do
      TR_SHADE_EXACT=$ex TINYRENDERER_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --contexts 1 --steps 4 --warmup 1 --frames-per-step 50 --no-cpu-baseline --no-secondary > /tmp/ab.json 2>/tmp/ab.err || { cat /tmp/ab.err; cat /tmp/ab.json; exit 1; }
      python3 -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print('$lib exact=$ex', 'verified', d['verified'], 'lsb', d['max_colour_lsb_vs_exact'], 'fps1', round(d['single_context_fps']), {k: round(v,1) for k,v in d['stages_us'].items() if v})"
    done
  done
done
