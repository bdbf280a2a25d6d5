#!/bin/bash
# Where the raster passes' time goes: SQ / LDS / TCP counters of the bench command for builds of the library,
# one counter set per run (through gpurun, from the repo root):
#   bash tools/pmc_raster.sh variants/a.so [variants/b.so ...]     -> gpurun_out/pmc_raster/<name>/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  OUT=gpurun_out/pmc_raster/$(basename "$lib" .so)
  rm -rf "$OUT" && mkdir -p "$OUT"
  BENCH="python3 bench.py --contexts 1 --steps 2 --warmup 1 --frames-per-step 10 --no-cpu-baseline --no-secondary --profile-frames 0"
  run() { name=$1; shift; TR_BENCH_NO_VERIFY=${NO_VERIFY:-0} TINYRENDERER_HIP_LIB=$PWD/$lib timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- $BENCH > "$OUT/$name.log" 2>&1 || { tail -5 "$OUT/$name.log"; exit 1; }; }
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES
  run sq2 SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
  run sq3 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_WAVE32_LDS
  run tcc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum
  run grbm GRBM_GUI_ACTIVE
  python3 - "$OUT" <<'PY' > $OUT/summary.txt
import csv, glob, collections, sys
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(out):
    if "k_raster" in k:
        print(k)
        for c, v in sorted(out[k].items()):
            print(f"   {c:45s} {sum(v)/len(v):16.1f}   (n={len(v)})")
PY
  echo "=== $lib"; cat $OUT/summary.txt
done
