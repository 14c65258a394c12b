#!/usr/bin/env python3
"""HBM / fabric read traffic per kernel of one UNet forward: one `rocprofv3 --pmc FETCH_SIZE` pass (counters only) over an eager
(GILL_NO_GRAPH=1) child of bench.py, grouped by (kernel, grid), printed as bytes per dispatch next to the dispatch count.

  python tools/pmc_by_kernel.py [--counter FETCH_SIZE|WRITE_SIZE] [--infer-steps 2] [--out gpurun_out/pmc_by_kernel.md]

FETCH_SIZE on gfx950 tallies 128-B requests as 64 B (MI355X_MICROARCH.md): the table applies the x2.  Both counters are in KiB.
"""
import argparse
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--counter", default="FETCH_SIZE")
  ap.add_argument("--infer-steps", type=int, default=2)
  ap.add_argument("--out", default=None)
  ap.add_argument("--top", type=int, default=60)
  ap.add_argument("--all", action="store_true", help="keep the model-construction kernels (weight conversion, OPT, mapper) in the table")
  a = ap.parse_args()
  from bench import SETUP_KERNELS
  rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
  d = tempfile.mkdtemp(prefix="gill_pmc_k_", dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp")
  env = dict(os.environ, GILL_NO_GRAPH="1", TMPDIR="/tmp")
  cmd = [rocprof, "--pmc", a.counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
         "--steps", "1", "--warmup", "0", "--child-pmc", "--infer-steps", str(a.infer_steps)]
  r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
  if r.returncode != 0:
    print(r.stdout.decode(errors="replace")[-2000:])
    sys.exit(r.returncode)
  agg = collections.OrderedDict()
  for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
      for row in csv.DictReader(fh):
        if row["Counter_Name"] != a.counter or (not a.all and (any(t in row["Kernel_Name"] for t in SETUP_KERNELS) or "at::native" in row["Kernel_Name"])):
          continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (name, row.get("Grid_Size", "?"), row.get("Workgroup_Size", "?"))
        e = agg.setdefault(key, [0, 0.0])
        e[0] += 1
        e[1] += float(row["Counter_Value"])
  shutil.rmtree(d, ignore_errors=True)
  scale = 2.0 if a.counter == "FETCH_SIZE" else 1.0
  rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
  total = sum(v[1] for v in agg.values()) * scale * 1024
  lines = [f"# {a.counter} per kernel ({a.infer_steps} loop steps; bytes = KiB x 1024" + (" x 2 (gfx950 FETCH_SIZE correction)" if scale == 2.0 else "") + ")",
           "", f"total {total / 1e9:.2f} GB", "", "| kernel | grid | wg | dispatches | MB per dispatch | total MB | share |", "|---|---|---|---|---|---|---|"]
  for (name, grid, wg), (n, kib) in rows[: a.top]:
    b = kib * scale * 1024
    lines.append(f"| `{name}` | {grid} | {wg} | {n} | {b / n / 1e6:.2f} | {b / 1e6:.1f} | {100 * b / total:.1f} % |")
  text = "\n".join(lines) + "\n"
  print(text)
  if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
      fh.write(text)


if __name__ == "__main__":
  main()
