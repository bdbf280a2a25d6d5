#!/bin/bash
# Where the batched orbit's kernels spend their time: SQ / LDS / atomic counters of tools/orbit_batch_bench.py, one counter
# set per run (through gpurun, from the repo root):   bash tools/pmc_orbit_sq.sh [lib.so]   -> gpurun_out/pmc_orbit_sq/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIB=${1:-tinyrenderer_amd/libtinyrenderer_hip.so}
OUT=gpurun_out/pmc_orbit_sq
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="python3 tools/orbit_batch_bench.py 128 32 1"
run() { name=$1; shift; TINYRENDERER_HIP_LIB=$PWD/$LIB timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- $CMD > "$OUT/$name.log" 2>&1 || { tail -5 "$OUT/$name.log"; exit 1; }; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES
run sq2 SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
run sq3 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_WAVE32_LDS
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
    if "tr::" in k and ("k_mid" in k or "k_shade" in k or "k_setup" in k):
        print(k)
        for c, v in sorted(out[k].items()):
            print(f"   {c:45s} {sum(v)/len(v):16.1f}   (n={len(v)})")
PY
cat $OUT/summary.txt
