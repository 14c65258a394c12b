"""MFMA-pipe busy fraction per kernel family from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass.
  python tools/pmc_mfma.py <dir> [out.md]
Collect with eager UNet forwards:  GILL_NO_GRAPH=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d <dir> -o m --output-format csv --
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc   (SQ counters over hipGraph replays did not finish within minutes on
ROCm 7.2: three attempts of 200 s each, with and without the ping-pong kernel; the eager run takes 7 s and launches the same kernels)
busy = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (1024 SIMDs x sum(GRBM_GUI_ACTIVE) / 8 XCDs)   (profiles/r01_pmc_conv_L0.md: the SQ counter is
in cycles summed over SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import csv, glob, sys, collections, re
busy = collections.defaultdict(float); act = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
  for row in csv.DictReader(open(f)):
    k = re.sub(r"^void ", "", row["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0][:48]
    v = float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": busy[k] += v; n[k] += 1
    elif row["Counter_Name"] == "GRBM_GUI_ACTIVE": act[k] += v
HOT = ("gemm_kernel", "attention_kernel", "attention_dma_kernel", "ffn_fused", "lnproj", "dup_pair", "gemm_splitk", "groupnorm", "conv_out", "conv3x3_fp8", "plms", "sd_stage", "im2col", "copy_bytes")
lines = ["| kernel | dispatches | GPU-active cycles (sum / 8 XCDs) | MFMA-busy cycles / (1024 SIMDs) | MFMA pipe busy |", "|---|---|---|---|---|"]
tb = ta = 0.0
for k in sorted(act, key=lambda k: -act[k]):
  if not any(h in k for h in HOT) or act[k] == 0: continue
  a = act[k] / 8; b = busy[k] / 1024
  tb += b; ta += a
  if a / max(ta, 1) > 0.002 or len(lines) < 22:
    lines.append(f"| `{k}` | {n[k]} | {a:.3e} | {b:.3e} | {100 * b / a:.1f} % |")
lines.append(f"| **all hot-path kernels** | | {ta:.3e} | {tb:.3e} | **{100 * tb / ta:.1f} %** |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2: open(sys.argv[2], "w").write("# MFMA pipe busy, whole bench step (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)\n\n" + out + "\n")
