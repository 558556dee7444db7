"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
