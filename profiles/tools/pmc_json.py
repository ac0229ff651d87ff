"""Summaries of the two counter passes of collect.sh:  pmc_json.py <dir with pmc_fetch.csv / pmc_write.csv> <out prefix>
-> <prefix>_pmc_k_gray_thresh3.json (what bench.py reads for roofline.traffic) and <prefix>_pmc_bytes_per_kernel.txt
(FETCH_SIZE / WRITE_SIZE of every kernel of the library, per launch).  Corrections as MI355X_MICROARCH.md prescribes:
both counters are in KiB; gfx950 tallies a 128-byte read request as 64 bytes, so FETCH_SIZE is doubled."""
import csv
import json
import sys
from collections import defaultdict

src, prefix = sys.argv[1], sys.argv[2]
H, W, FRAMES = 1080, 1920, 4096


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("camsync::(anonymous namespace)::", "").replace("void ", "")
        name = name[:name.index("(")] if "(" in name else name
        tot[name] += float(r["Counter_Value"])
        n[name] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


f = per_kernel(src + "/pmc_fetch.csv", "FETCH_SIZE")
w = per_kernel(src + "/pmc_write.csv", "WRITE_SIZE")
with open(prefix + "_pmc_bytes_per_kernel.txt", "w") as out:
    out.write("# HBM bytes per launch: FETCH_SIZE x 1024 x 2 (gfx950 correction), WRITE_SIZE x 1024; one bench step of 4096 1080p frames\n")
    out.write("%-34s %8s %14s %14s\n" % ("kernel", "launches", "fetch MB", "write MB"))
    for k in f:
        out.write("%-34s %8d %14.1f %14.1f\n" % (k, f[k][1], f[k][0] * 2048 / 1e6, w.get(k, (0, 0))[0] * 1024 / 1e6))
k = [x for x in f if x.startswith("k_gray_thresh3")][0]
px = FRAMES * H * W
fb, wb = f[k][0] * 2048, w[k][0] * 1024
json.dump({
    "kernel": k,
    "config": {"frames_per_launch": FRAMES, "height": H, "width": W, "channels": 3},
    "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --calib-views 0   (profiles/tools/collect.sh)",
    "raw": {"FETCH_SIZE_KiB_per_launch": f[k][0], "WRITE_SIZE_KiB_per_launch": w[k][0], "launches": [f[k][1], w[k][1]]},
    "corrections": "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE taken as is; KiB = 1024 B.  Expected: reads 3 B/px x 512/480 strip halo = 3.2 B/px; writes 3/8 B/px of bit planes (the pipeline instance stores no gray image).",
    "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb,
    "fetch_bytes_per_px": fb / px, "write_bytes_per_px": wb / px, "traffic_bytes_per_px": (fb + wb) / px,
    "algorithmic_bytes_per_px": 8.0,
}, open(prefix + "_pmc_k_gray_thresh3.json", "w"), indent=1)
