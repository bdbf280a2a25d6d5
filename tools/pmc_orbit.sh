cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_orbit; rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum --output-format csv -d $out -o pmc -- python3 tools/orbit_batch_bench.py 128 32 1 > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - "$(ls $out/*/pmc_counter_collection.csv $out/pmc_counter_collection.csv 2>/dev/null | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "tr::" in k: acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/sq -o pmc -- python3 tools/orbit_batch_bench.py 128 32 1 > $out/log2.txt 2>&1 || { tail -5 $out/log2.txt; exit 1; }
python3 - "$(ls $out/sq/*/pmc_counter_collection.csv $out/sq/pmc_counter_collection.csv 2>/dev/null | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "tr::" in k: acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
