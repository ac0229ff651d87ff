"""Short table of a rocprofv3 kernel_stats.csv: name (trimmed), calls, average and total microseconds."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
only = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    n = r["Name"]
    if only and only not in n:
        continue
    short = re.sub(r"\(.*", "", n).replace("void ", "").replace("camsync::(anonymous namespace)::", "")
    print("%-60s calls %6s avg %10.1f us total %10.1f us" % (short[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
