#!/bin/bash
# One SQ counter pass:  bash profiles/tools/pmc_sq.sh <tag> [kernel substring] [-- program args...]
# Default program: one bench step.  (8 SQ counters is the most one pass can hold.)  Prints per-kernel sums per launch.
set -e
tag=${1:-sq}
sub=$2
shift; shift || true
if [ "$1" = "--" ]; then shift; else set -- bench.py --steps 1 --warmup 0 --no-cpu --no-extra --calib-views 0; fi
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d /tmp/pmc_sq -- python3 "$@" > "$out/pmc_sq.log" 2>&1
python3 profiles/tools/pmc_sum.py /tmp/pmc_sq $sub > "$out/pmc_sq.txt"
rm -rf /tmp/pmc_sq
cat "$out/pmc_sq.txt"
