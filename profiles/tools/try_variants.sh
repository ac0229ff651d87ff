#!/bin/bash
# Developer aid: time the bench's stages with each library variant under variants/*.so (built with different constants).
# usage (on the GPU box): [BENCH_ARGS="--frames 256 --chunk 256"] bash profiles/tools/try_variants.sh
set -e
lib=cameracalibration_amd/libcamsync_hip.so
cp $lib /tmp/lib_orig.so
for v in variants/*.so; do
  cp "$v" $lib
  timeout -k 10 120 python3 bench.py --no-cpu --calib-views 0 --steps 3 --warmup 1 $BENCH_ARGS > /tmp/b.json
  python3 -c "
import json,sys;d=json.load(open('/tmp/b.json'));s=d['stage_ms_per_step'];print('$v', round(d['value']), 'thresh', s['gray_threshold'], 'starts', s['contour_starts'], 'trace', s['trace'], 'total', s['total'], 'poses', d['check']['frames_with_pose'])"
done
cp /tmp/lib_orig.so $lib
