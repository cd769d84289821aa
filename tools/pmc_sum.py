"""Sums rocprofv3 --pmc counters per kernel name prefix: python tools/pmc_sum.py <dir> <kernel prefix>"""
import collections, csv, glob, sys
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(sys.argv[2]):
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print("%-24s %.6g" % (k, v))
