#!/usr/bin/env python3
"""gpurun_out/prof (tools/collect_profiles.sh) -> profiles/<tag>_kernel_stats_4k.csv,
profiles/<tag>_hbm_pmc_4k.json, profiles/<tag>_bench_4k.json."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)


def counters(sub):
    """mean counter value per launch, per kernel (short name)"""
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats_4k.csv")
stats_exact = glob.glob(os.path.join(src, "stats_exact", "**", "*kernel_stats.csv"), recursive=True)
if stats_exact:
    shutil.copy(stats_exact[0], f"profiles/{tag}_kernel_stats_4k_exact_colour.csv")
stats4 = glob.glob(os.path.join(src, "stats4", "**", "*kernel_stats.csv"), recursive=True)
if stats4:
    shutil.copy(stats4[0], f"profiles/{tag}_kernel_stats_4k_4ctx.csv")
if os.path.exists(os.path.join(src, "overlap4.txt")):
    shutil.copy(os.path.join(src, "overlap4.txt"), f"profiles/{tag}_overlap_4k_4ctx.txt")
for sub, name in (("orbit", "orbit_1080p_1ctx"), ("orbit_batch", "orbit_1080p_batch32"), ("band", "band_balanced_rows2288_2592_10M_8k"),
                  ("shadow", "diablo_shadow_1024"), ("bigmesh", "bigmesh_10M_8k")):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], f"profiles/{tag}_kernel_stats_{name}.csv")
pmc = {"source": "tools/collect_profiles.sh: rocprofv3 --pmc, one counter set per run, python3 bench.py --contexts 1 "
                 "--steps 2 --warmup 1 --frames-per-step 20, african_head 3840x2160, global raster, RGBA8-only resolve",
       "units": "FETCH_SIZE / WRITE_SIZE in KB per launch as rocprofv3 reports them, averaged over launches",
       "gfx950_correction": "FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): doubled; WRITE_SIZE as is",
       "workload": {"W": 3840, "H": 2160, "scene": "african_head"}, "kernels": {}}
sets = {s: counters(s) for s in ("fetch", "write", "sq", "tcc", "ta", "tcp")}
names = set()
for s in sets.values():
    names |= set(s)
for k in sorted(names):
    e = {}
    for s in sets.values():
        e.update(s.get(k, {}))
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
    pmc["kernels"][k] = e
def shade_kernel(kernels):
    """the kernel the timed frames run: k_shade_fused unless TR_SHADE_EXACT=1 (bench.py's check re-shades once with k_shade)"""
    for prefix in ("tr::k_shade_fused", "tr::k_shade"):
        for k, v in kernels.items():
            if k.startswith(prefix):
                return k, v
    return "", {}


shade_name, sh = shade_kernel(pmc["kernels"])
pmc["k_shade_kernel"] = shade_name
if "hbm_bytes_per_launch" in sh:
    pmc["k_shade_hbm_bytes_per_launch"] = sh["hbm_bytes_per_launch"]
if "SQ_INSTS_VALU" in sh:
    pmc["k_shade_valu_insts_per_launch"] = sh["SQ_INSTS_VALU"]
# the raster pair (the larger part of the frame): what bench.py's `raster_passes` record quotes beside its live stage times
for short, needle in (("k_depth", "k_raster_depth<256"), ("k_winner", "k_raster_winner<256")):
    for k, e in pmc["kernels"].items():
        if needle in k:
            pmc[short + "_kernel"] = k
            for c, name in (("SQ_INSTS_VALU", "valu_insts_per_launch"), ("TCC_EA0_ATOMIC_sum", "atomic_requests_per_launch"),
                            ("hbm_bytes_per_launch", "hbm_bytes_per_launch")):
                if c in e:
                    pmc[f"{short}_{name}"] = e[c]
json.dump(pmc, open(f"profiles/{tag}_hbm_pmc_4k.json", "w"), indent=1)
# the same counters with four contexts in flight (working set 4 x 166 MB > the 256 MB Infinity Cache)
pmc4 = {"source": "tools/collect_profiles.sh: rocprofv3 --pmc, one counter set per run, python3 bench.py (FOUR contexts in flight) "
                  "--steps 2 --warmup 1 --frames-per-step 20, african_head 3840x2160, global raster, RGBA8-only resolve",
        "units": pmc["units"], "gfx950_correction": pmc["gfx950_correction"], "workload": dict(pmc["workload"], contexts=4), "kernels": {}}
sets4 = {s: counters(s) for s in ("fetch4", "write4", "tcc4")}
names4 = set()
for s_ in sets4.values():
    names4 |= set(s_)
for k in sorted(names4):
    e = {}
    for s_ in sets4.values():
        e.update(s_.get(k, {}))
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
    pmc4["kernels"][k] = e
_, sh4 = shade_kernel(pmc4["kernels"])
if "hbm_bytes_per_launch" in sh4:
    pmc4["k_shade_hbm_bytes_per_launch"] = sh4["hbm_bytes_per_launch"]
if pmc4["kernels"]:
    json.dump(pmc4, open(f"profiles/{tag}_hbm_pmc_4k_4ctx.json", "w"), indent=1)
b = os.path.join(src, "bench.json")
if os.path.exists(b):
    lines = [l for l in open(b) if l.startswith("{")]
    if lines:
        line = json.loads(lines[-1])
        if line.get("roofline", {}).get("traffic") is None and "k_shade_hbm_bytes_per_launch" in pmc:
            line["roofline"]["traffic"] = pmc["k_shade_hbm_bytes_per_launch"]  # (bench.py ran before this file existed)
        open(f"profiles/{tag}_bench_4k.json", "w").write(json.dumps(line) + "\n")
print(json.dumps(pmc, indent=1)[:3000])
