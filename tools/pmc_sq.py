"""Per-kernel SQ counter table from rocprofv3 --pmc passes over eager UNet forwards (GILL_NO_GRAPH=1): where do the waves of each
kernel spend their cycles?   python tools/pmc_sq.py <dir with the passes' csv files> [filter substring]
Rows: kernel (template arguments kept) x grid; columns: every counter found, summed over the kernel's dispatches and divided by
SQ_WAVE_CYCLES of the same dispatches where that is the natural unit (WAIT_*, ACTIVE_INST_*), else raw per dispatch."""
import csv, glob, sys, collections, re
val = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float); regs = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
  for row in csv.DictReader(open(f)):
    name = re.sub(r"^void ", "", row["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0]
    k = (name[:52], int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1))
    val[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
    regs[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("LDS_Block_Size"))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = sorted({c for k in val for c in val[k]})
print("kernel | WGs | n | vgpr/agpr/lds | " + " | ".join(n.replace("SQ_", "") for n in names))
for k in sorted(val, key=lambda k: -val[k].get("SQ_WAVE_CYCLES", 0)):
  if flt not in k[0]: continue
  wc = val[k].get("SQ_WAVE_CYCLES", 0) / max(cnt[k].get("SQ_WAVE_CYCLES", 1), 1)
  cells = []
  for n in names:
    v = val[k][n] / max(cnt[k][n], 1)
    if wc and (n.startswith("SQ_WAIT") or n.startswith("SQ_ACTIVE_INST") or n == "SQ_INST_CYCLES_SALU"): cells.append(f"{100 * v / wc:.1f}%")
    else: cells.append(f"{v:.3e}")
  print(f"{k[0]} | {k[1]} | {max(cnt[k].values())} | {'/'.join(str(r) for r in regs[k])} | " + " | ".join(cells))
