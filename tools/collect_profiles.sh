#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: kernel-time statistics and HBM / VALU
# counters of the bench command, each counter set in a run of its own (rocprofv3 does not split
# counters over passes).  Output under gpurun_out/prof/; tools/summarise_profiles.py turns it into
# the files committed under profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
rm -rf "$OUT" && mkdir -p "$OUT"
BENCH="python3 bench.py --contexts 1 --steps 2 --warmup 1 --frames-per-step 20 --no-cpu-baseline --no-secondary --profile-frames 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o ks -- $BENCH > "$OUT/stats.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o pmc -- $BENCH > "$OUT/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o pmc -- $BENCH > "$OUT/write.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -o pmc -- $BENCH > "$OUT/sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -o pmc -- $BENCH > "$OUT/tcc.log" 2>&1
# HBM bytes with FOUR contexts in flight (the headline command's mode: 4 x 166 MB of planes, past the 256 MB Infinity Cache)
BENCH4="python3 bench.py --steps 2 --warmup 1 --frames-per-step 20 --no-cpu-baseline --no-secondary --profile-frames 0"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch4" -o pmc -- $BENCH4 > "$OUT/fetch4.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write4" -o pmc -- $BENCH4 > "$OUT/write4.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc4" -o pmc -- $BENCH4 > "$OUT/tcc4.log" 2>&1
# what bounds the shade pass: texture-unit / L1 counters (tools/pmc_shade.sh holds the full set)
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum --output-format csv -d "$OUT/ta" -o pmc -- $BENCH > "$OUT/ta.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/tcp" -o pmc -- $BENCH > "$OUT/tcp.log" 2>&1
# the exact colour arithmetic (TR_SHADE_EXACT=1), kernels one at a time: the A/B of DESIGN.md 4.4
TR_SHADE_EXACT=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_exact" -o ks -- $BENCH > "$OUT/stats_exact.log" 2>&1
# the headline command itself (four contexts in flight): kernel durations there include the time a kernel
# shares the GPU with those of three other frames; tools/overlap_report.py summarises the overlap
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats4" -o ks -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --profile-frames 0 > "$OUT/stats4.log" 2>&1
python3 tools/overlap_report.py "$(ls $OUT/stats4/*/ks_kernel_trace.csv $OUT/stats4/ks_kernel_trace.csv 2>/dev/null | head -1)" > "$OUT/overlap4.txt" 2>&1 || true
# the other configurations of BASELINE.json rehearsed on one GPU: kernel summaries only
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/orbit" -o ks -- python3 tools/orbit_bench.py 512 1920 1080 1 > "$OUT/orbit.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/orbit_batch" -o ks -- python3 tools/orbit_batch_bench.py 512 32 1 > "$OUT/orbit_batch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/band" -o ks -- python3 tools/band_one.py 3 8 6 6 2288 2592 > "$OUT/band.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/shadow" -o ks -- python3 tests/manual/diablo_shadow.py 1024 1024 1024 > "$OUT/shadow.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bigmesh" -o ks -- python3 tests/manual/big_mesh_check.py 6 7680 4320 0 0 > "$OUT/bigmesh.log" 2>&1
# the headline command without the profiler, for the committed bench line
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
ls -R "$OUT" | head -40
