#!/bin/bash
# One SQ counter pass over one bench step:  bash profiles/tools/pmc_sq.sh <tag> [kernel substring]
# (8 SQ counters is the most one pass can hold.)  Prints per-kernel sums per launch.
set -e
tag=${1:-sq}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d /tmp/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu --calib-views 0 > "$out/pmc_sq.log" 2>&1
python3 profiles/tools/pmc_sum.py /tmp/pmc_sq $2 > "$out/pmc_sq.txt"
rm -rf /tmp/pmc_sq
cat "$out/pmc_sq.txt"
