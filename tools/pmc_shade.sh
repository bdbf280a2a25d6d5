#!/bin/bash
# Where k_shade's time goes: SQ / TA / TCP / TD counters of the bench command, one counter set per run.
#   gpurun -- bash tools/pmc_shade.sh     -> gpurun_out/pmc_shade/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_shade
rm -rf "$OUT" && mkdir -p "$OUT"
BENCH="python3 bench.py --contexts 1 --steps 2 --warmup 1 --frames-per-step 20 --no-cpu-baseline --no-secondary --profile-frames 0"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- $BENCH > "$OUT/$name.log" 2>&1 || { tail -5 "$OUT/$name.log"; exit 1; }; }
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
run ta TA_BUSY_avr TA_BUSY_max
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM
run tcp TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
run ta2 TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run td TD_TD_BUSY_sum TD_TC_STALL_sum
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_shade/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(out):
    if "k_shade" in k or "k_raster" in k:
        print(k)
        for c, v in sorted(out[k].items()):
            print(f"   {c:45s} {sum(v)/len(v):16.1f}   (n={len(v)})")
PY
cat $OUT/summary.txt
