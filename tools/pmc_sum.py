"""per-kernel sums of one rocprofv3 counter_collection.csv:  pmc_sum.py <csv> <counter> <name-substring>..."""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if any(p in k for p in sys.argv[3:]):
        m = re.search(r"(k_\w+)(<[^>(]*>)?", k)
        name = (m.group(1) + (m.group(2) or "")) if m else k[:60]
        acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
for k, (v, n) in sorted(acc.items()):
    print(sys.argv[2], k, "sum", v, "launches", n, "per_launch", v / n)
