"""Keep only this library's kernels from a rocprofv3 counter_collection CSV (run on the GPU box: the raw file
exceeds what gpurun merges back).  usage: filt.py <rocprof output dir> <out.csv>"""
import csv
import glob
import sys

for p in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    rows = list(csv.reader(open(p)))
    hdr = rows[0]
    ki = hdr.index("Kernel_Name")
    csv.writer(open(sys.argv[2], "w", newline="")).writerows([hdr] + [r for r in rows[1:] if "camsync" in r[ki]])
