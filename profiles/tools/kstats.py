"""Short table of a rocprofv3 kernel_stats.csv: kernel name (trimmed), calls, average and total microseconds."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
only = sys.argv[2] if len(sys.argv) > 2 else "camsync"
for r in rows:
    n = r["Name"]
    if only and only not in n:
        continue
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    print("%-34s calls %6s avg %10.1f us total %11.1f us" % (m.group(1) if m else n[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
