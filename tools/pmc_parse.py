import csv, sys, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] not in row["Kernel_Name"]:
            continue
        a = agg[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f"{k:32s} {v / n:14.4e}  (n={n})")
