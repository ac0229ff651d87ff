#!/bin/bash
# Developer aid: FETCH_SIZE of one kernel for each library variant under variants/*.so.  usage: try_variants_fetch.sh <kernel substring>
set -e
lib=cameracalibration_amd/libcamsync_hip.so
cp $lib /tmp/lib_orig.so
export TMPDIR=/tmp
for v in variants/*.so; do
  cp "$v" $lib
  rm -rf /tmp/pf
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 bench.py --steps 1 --warmup 0 --no-cpu --calib-views 0 > /dev/null 2>&1
  echo "$v $(python3 profiles/tools/pmc_sum.py /tmp/pf $1)"
done
cp /tmp/lib_orig.so $lib
