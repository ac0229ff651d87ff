#!/bin/bash
# Developer aid: run one leg of profiles/tools/leg.py with each library variant under variants/*.so and print chosen keys.
# usage (on the GPU box): bash profiles/tools/try_variants_leg.sh mjpeg frames_per_s ms_per_batch
leg=$1; shift
lib=cameracalibration_amd/libcamsync_hip.so
cp $lib /tmp/lib_orig.so
for v in /tmp/lib_orig.so variants/*.so; do
  [ "$v" != /tmp/lib_orig.so ] && cp "$v" $lib
  timeout -k 10 200 python3 profiles/tools/leg.py $leg > /tmp/b.json 2>/dev/null || { echo "$v FAILED"; continue; }
  python3 -c "
import json,sys;d=json.load(open('/tmp/b.json'));print('$v', {k:d.get(k) for k in sys.argv[1:]})" "$@"
done
cp /tmp/lib_orig.so $lib
