#!/bin/bash
# Iteration loop on the GPU box:  bash profiles/tools/quick.sh <tag> [pytest -k expression | none]
# the GPU tests (all, or the ones selected), then the per-stage times of the bench scene, the photo scene and the 4K scene.
tag=${1:-q}
sel=$2
out=gpurun_out/$tag
mkdir -p $out
if [ "$sel" != "none" ]; then
  if [ -n "$sel" ]; then
    timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "$sel" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  else
    timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  fi
  tail -1 $out/tests.log
fi
timeout -k 10 200 python3 bench.py --no-cpu --no-extra --calib-views 0 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 120 python3 profiles/tools/leg.py photo > $out/photo.json 2> $out/photo.err || { tail -5 $out/photo.err; exit 1; }
timeout -k 10 120 python3 profiles/tools/leg.py 4k > $out/4k.json 2> $out/4k.err || { tail -5 $out/4k.err; exit 1; }
python3 - <<PY
import json
b = json.load(open("$out/bench.json"))
print("bench %.0f f/s" % b["value"], b["stage_ms_per_step"], "ok", b["check"]["frames_with_pose"])
for n in ("photo", "4k"):
    d = json.load(open("$out/%s.json" % n))
    print(n, "%.0f f/s" % d["frames_per_s"], d["stage_ms_per_step"])
PY
