#!/bin/bash
# Memory-side atomic requests and instruction counts of the 10^7-triangle frame's kernels (through gpurun, from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_bigmesh; rm -rf $out; mkdir -p $out
for set in "TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  d=$out/$(echo $set | cut -c1-6); mkdir -p $d
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $d -o pmc -- python3 tests/manual/big_mesh_check.py 6 7680 4320 0 0 > $d/log.txt 2>&1 || { tail -5 $d/log.txt; exit 1; }
  python3 - "$(ls $d/*/pmc_counter_collection.csv $d/pmc_counter_collection.csv 2>/dev/null | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "tr::" in k: acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if max(max(v) for v in d.values()) > 1e5: print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
