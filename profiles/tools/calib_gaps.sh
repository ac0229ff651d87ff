#!/bin/bash
# Durations of the LM kernels of one calibration and the gaps between them (us):  bash profiles/tools/calib_gaps.sh
export TMPDIR=/tmp
timeout -k 10 100 python3 profiles/tools/calib_only.py
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ck_tmp -- python3 profiles/tools/calib_only.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/ck_tmp/*/*kernel_trace.csv")[0])))
rows=[r for r in rows if "k_calib_lm<12>" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
prev=None
out=[]
for r in rows[-30:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    out.append((round((s-prev)/1e3,1) if prev else None, round((e-s)/1e3,1)))
    prev=e
print(len(rows), "launches; (gap before, duration) of the last 30:", out)
PY
rm -rf gpurun_out/ck_tmp
