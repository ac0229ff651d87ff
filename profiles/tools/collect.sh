#!/bin/bash
# Collect the round's profiles on the GPU box:  bash profiles/tools/collect.sh <tag>
# 1. the default bench line, 2. rocprofv3 kernel stats of the same command, 3./4. FETCH_SIZE and WRITE_SIZE of the
# threshold kernel in separate counter passes (one step, no warm-up, no CPU leg).  Outputs under gpurun_out/<tag>/.
set -e
tag=${1:-prof}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 280 python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --no-cpu --no-extra > "$out/kt.log" 2>&1
cp "$out"/kt/*/*kernel_stats.csv "$out/kernel_stats.csv"
rm -rf "$out/kt"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --calib-views 0 > "$out/pmc_f.log" 2>&1
python3 profiles/tools/filt.py /tmp/pmc_f "$out/pmc_fetch.csv"
rm -rf /tmp/pmc_f
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --calib-views 0 > "$out/pmc_w.log" 2>&1
python3 profiles/tools/filt.py /tmp/pmc_w "$out/pmc_write.csv"
rm -rf /tmp/pmc_w
ls -la "$out"
