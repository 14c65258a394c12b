"""Sum a PMC counter over every dispatch of a rocprofv3 --pmc CSV run, split by kernel family.
  python tools/pmc_sum.py <dir> <counter>"""
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != sys.argv[2]:
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        tot[k] += float(row["Counter_Value"]); n[k] += 1
SETUP = ("at::native", "__amd_rocclr", "convert_", "relayout", "pad_head", "scatter_rows", "permute", "ln_fold", "vec_add", "cast_")
allv = sum(tot.values())
path = sum(v for k, v in tot.items() if not any(t in k for t in SETUP))
print(f"{sys.argv[2]} hot-path kernels only (weight set-up / torch kernels excluded): {path:.4e}")
print(f"{sys.argv[2]} total {allv:.4e} over {sum(n.values())} dispatches")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {k:60s} {v:12.4e}  n={n[k]}")
