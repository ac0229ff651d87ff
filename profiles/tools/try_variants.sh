#!/bin/bash
# Developer aid: time the stages of the bench scene / a leg with each library variant under variants/*.so.
# usage (on the GPU box): [WHAT="bench photo 4k"] bash profiles/tools/try_variants.sh
lib=cameracalibration_amd/libcamsync_hip.so
cp $lib /tmp/lib_orig.so
for v in /tmp/lib_orig.so variants/*.so; do
  [ "$v" != /tmp/lib_orig.so ] && cp "$v" $lib
  for w in ${WHAT:-bench}; do
    if [ $w = bench ]; then
      timeout -k 10 120 python3 bench.py --no-cpu --no-extra --calib-views 0 --steps 3 --warmup 1 $BENCH_ARGS > /tmp/b.json 2>/dev/null || { echo "$v $w FAILED"; continue; }
      python3 -c "
import json;d=json.load(open('/tmp/b.json'));print('$v', 'bench', round(d['value']), d['stage_ms_per_step'], 'poses', d['check']['frames_with_pose'])"
    else
      timeout -k 10 120 python3 profiles/tools/leg.py $w > /tmp/b.json 2>/dev/null || { echo "$v $w FAILED"; continue; }
      python3 -c "
import json;d=json.load(open('/tmp/b.json'));print('$v', '$w', round(d['frames_per_s']), d['stage_ms_per_step'], 'markers', d['mean_markers_per_frame'])"
    fi
  done
done
cp /tmp/lib_orig.so $lib
