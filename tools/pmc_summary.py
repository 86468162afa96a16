#!/usr/bin/env python3
"""Average PMC counter values per kernel family from a rocprofv3 counter_collection.csv.  usage: pmc_summary.py csv substr..."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for sub in sys.argv[2:]:
    agg = collections.defaultdict(list)
    d = []
    for r in rows:
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    n = max(len(v) for v in agg.values()) if agg else 0
    print(sub, "launches", n, {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
