#!/bin/bash
# Kernel stats of the photo and the 4K leg:  bash profiles/tools/legs_stats.sh <tag>  ->  gpurun_out/<tag>/{photo,4k}_kernel_stats.csv
set -e
tag=${1:-legs}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for w in photo 4k; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_$w" -- python3 profiles/tools/leg.py $w > "$out/$w.log" 2>&1
  cp "$out"/kt_$w/*/*kernel_stats.csv "$out/${w}_kernel_stats.csv"
  rm -rf "$out/kt_$w"
done
ls -la "$out"
