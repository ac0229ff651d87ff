#!/bin/bash
# Developer aid: the bench scene's stage times with each library variant under variants/*.so
lib=cameracalibration_amd/libcamsync_hip.so
cp $lib /tmp/lib_orig.so
for v in /tmp/lib_orig.so variants/*.so; do
  [ "$v" != /tmp/lib_orig.so ] && cp "$v" $lib
  timeout -k 10 200 python3 bench.py --no-cpu --no-extra --calib-views 0 > /tmp/b.json 2>/dev/null || { echo "$v bench FAILED"; break; }
  python3 -c "
import json,sys;b=json.load(open('/tmp/b.json'));print('$v', '%.0f' % b['value'], b['stage_ms_per_step'], b['check']['frames_with_pose'])"
done
cp /tmp/lib_orig.so $lib
