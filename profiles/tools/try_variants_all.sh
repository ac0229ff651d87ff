#!/bin/bash
# Developer aid: the bench scene, the photo scene and the 4K scene with each library variant under variants/*.so
# (stage times of the border stage and the whole step).  usage (on the GPU box): bash profiles/tools/try_variants_all.sh
lib=cameracalibration_amd/libcamsync_hip.so
cp $lib /tmp/lib_orig.so
for v in /tmp/lib_orig.so variants/*.so; do
  [ "$v" != /tmp/lib_orig.so ] && cp "$v" $lib
  timeout -k 10 200 python3 bench.py --no-cpu --no-extra --calib-views 0 > /tmp/b.json 2>/dev/null || { echo "$v bench FAILED"; break; }
  timeout -k 10 120 python3 profiles/tools/leg.py photo > /tmp/p.json 2>/dev/null || { echo "$v photo FAILED"; break; }
  timeout -k 10 120 python3 profiles/tools/leg.py 4k > /tmp/k.json 2>/dev/null || { echo "$v 4k FAILED"; break; }
  python3 - "$v" <<PY
import json, sys
b = json.load(open("/tmp/b.json")); p = json.load(open("/tmp/p.json")); k = json.load(open("/tmp/k.json"))
def st(d): s = d["stage_ms_per_step"]; return "starts %.2f trace %.2f quad %.2f total %.2f" % (s.get("contour_starts", 0), s.get("trace", 0), s.get("quad_fit", 0), sum(s.values()))
print(sys.argv[1], "| bench %.0f" % b["value"], st(b), "ok", b["check"]["frames_with_pose"], "| photo %.0f" % p["frames_per_s"], st(p), "| 4k %.0f" % k["frames_per_s"], st(k))
PY
done
cp /tmp/lib_orig.so $lib
