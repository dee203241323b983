"""Aggregates a rocprofv3 --pmc counter_collection csv per kernel: mean counter value per dispatch."""
import csv, sys, collections, glob
files = []
for a in sys.argv[1:]:
    files += glob.glob(a + "/**/*counter_collection.csv", recursive=True) if not a.endswith(".csv") else [a]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f"   {c:32s} {acc[k][c] / cnt[k][c]:16.1f}  (x{cnt[k][c]})")
