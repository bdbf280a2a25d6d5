#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 5 60 rocprofv3 -L > gpurun_out/pmc_list.txt 2>&1
grep -c . gpurun_out/pmc_list.txt
