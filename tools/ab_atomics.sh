#!/bin/bash
# Memory-side atomic requests per launch of the raster kernels, for builds of the library (through gpurun, from the repo root):
#   bash tools/ab_atomics.sh variants/a.so variants/b.so [...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  out=gpurun_out/ab_atomics/$(basename "$lib" .so)
  rm -rf "$out" && mkdir -p "$out"
  TR_BENCH_NO_VERIFY=${NO_VERIFY:-0} TINYRENDERER_HIP_LIB=$PWD/$lib timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum --output-format csv -d "$out" -o pmc -- python3 bench.py --contexts 1 --steps 2 --warmup 1 --frames-per-step 10 --no-cpu-baseline --no-secondary --profile-frames 0 > "$out/log.txt" 2>&1 || { tail -5 "$out/log.txt"; exit 1; }
  python3 - "$lib" "$(ls $out/*/pmc_counter_collection.csv $out/pmc_counter_collection.csv 2>/dev/null | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[2])):
    k = r["Kernel_Name"]
    if "k_raster" in k or "k_shade" in k or "k_mid" in k or "k_small" in k:
        acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(sys.argv[1], k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
