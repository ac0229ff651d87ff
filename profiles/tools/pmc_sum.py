"""Per-kernel sums of a rocprofv3 --pmc counter_collection CSV.  usage: pmc_sum.py <rocprof output dir> [kernel substring]"""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "camsync" not in n or (len(sys.argv) > 2 and sys.argv[2] not in n):
            continue
        m = re.search(r"(k_\w+(<[^>]*>)?)", n)
        k = m.group(1) if m else n[:30]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "launches", len(calls[k]), {c: "%.4g" % (v / len(calls[k])) for c, v in sorted(acc[k].items())})
