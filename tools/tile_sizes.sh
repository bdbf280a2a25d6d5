#!/bin/bash
# Tile raster (TR_RASTER_MODE=tile) at every tile size k_tile_raster<TW, TH> is instantiated for: parity tests, then the
# 4K frame's per-stage times on one context (stage "depth_pass" = binning, "winner_pass" = the tile raster in this mode).
#   gpurun -- bash tools/tile_sizes.sh        -> gpurun_out/tile_sizes.txt
set -o pipefail
out=gpurun_out/tile_sizes.txt
mkdir -p gpurun_out
: > $out
for t in 16x16 32x16 32x32 64x32; do
  echo "== TR_TILE=$t" >> $out
  TR_TILE=$t timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 >> $out || exit 1
  TR_BENCH_NO_VERIFY=1 TR_TILE=$t TR_RASTER_MODE=tile timeout -k 10 120 python bench.py --no-secondary --no-cpu-baseline --contexts 1 --steps 4 --warmup 2 --frames-per-step 50 2>>$out | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('verified',d['verified'],'fps',round(d['value']),'stages',{k:round(v,1) for k,v in d['stages_us'].items()})" >> $out || exit 1
done
cat $out
