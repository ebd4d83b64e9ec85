"""Aggregate rocprofv3 --pmc CSV output: average counter value per launch for kernels matching a substring."""
import collections
import csv
import glob
import sys

root, needle = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if needle in k:
            agg[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k[:60]][r["Counter_Name"]] += 1
for k, v in agg.items():
    print(k)
    for c, s in sorted(v.items()):
        print("   %-28s %14.0f  (launches %d)" % (c, s / cnt[k][c], cnt[k][c]))
