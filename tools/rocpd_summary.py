"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel table:
  python tools/rocpd_summary.py gpurun_out/prof/bench_results.db [out.md]
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
  name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
  name = name.replace("void ", "")
  return name[:110]


def main():
  db = sqlite3.connect(sys.argv[1])
  cur = db.cursor()
  cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
  # the `kernels` view carries name / start / end per dispatch
  namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
  rows = cur.execute(f"select {namecol}, start, end from kernels").fetchall()
  agg = {}
  t_min, t_max = min(r[1] for r in rows), max(r[2] for r in rows)
  for n, s, e in rows:
    a = agg.setdefault(short(n), [0, 0, 10 ** 18, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
  total = sum(a[1] for a in agg.values())
  lines = [f"# rocprofv3 kernel summary: {sys.argv[1]}", "",
           f"dispatches {len(rows)}, summed kernel time {total / 1e6:.2f} ms, first-start to last-end span {(t_max - t_min) / 1e6:.2f} ms", "",
           "| kernel | calls | total ms | % | avg us | min us | max us |", "|---|---|---|---|---|---|---|"]
  for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"| `{k}` | {a[0]} | {a[1] / 1e6:.2f} | {100.0 * a[1] / total:.1f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} |")
  # inter-kernel gaps (start[i+1] - end[i] in start order); gaps > 200 us are host pauses, not dispatch cost
  rs = sorted(rows, key=lambda r: r[1])
  gaps = [(rs[i + 1][1] - rs[i][2], short(rs[i][0])) for i in range(len(rs) - 1)]
  small = [g for g, _ in gaps if 0 <= g < 200e3]
  overl = sum(1 for g, _ in gaps if g < 0)
  if small:
    ss = sorted(small)
    lines += ["", f"inter-kernel gaps < 200 us: n={len(small)}, sum {sum(small) / 1e6:.2f} ms, median {ss[len(ss) // 2] / 1e3:.2f} us, "
                  f"p90 {ss[int(len(ss) * 0.9)] / 1e3:.2f} us, mean {sum(small) / len(small) / 1e3:.2f} us; overlapping pairs {overl}"]
    by = {}
    for g, n in gaps:
      if 0 <= g < 200e3:
        b_ = by.setdefault(n, [0, 0]); b_[0] += 1; b_[1] += g
    lines += ["", "| gap AFTER kernel | n | mean gap us | total ms |", "|---|---|---|---|"]
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
      lines.append(f"| `{k}` | {v[0]} | {v[1] / v[0] / 1e3:.2f} | {v[1] / 1e6:.2f} |")
  out = "\n".join(lines) + "\n"
  if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
  print(out)


if __name__ == "__main__":
  main()


def per_shape(dbpath, out=None):
  """Break the GEMM / attention / GroupNorm kernels down by launch grid (one grid == one layer shape)."""
  db = sqlite3.connect(dbpath)
  cur = db.cursor()
  rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels").fetchall()
  agg = {}
  for n, s, e, gx, gy, gz, wx in rows:
    k = short(n)
    if not any(t in k for t in ("gemm_", "attention_", "groupnorm_", "conv_out")):
      continue
    key = (k, gx // max(wx, 1), gy, gz)
    a = agg.setdefault(key, [0, 0])
    a[0] += 1; a[1] += e - s
  lines = ["", "## per-shape breakdown (grid in workgroups)", "", "| kernel | grid | calls | total ms | avg us |", "|---|---|---|---|---|"]
  for (k, gx, gy, gz), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
    lines.append(f"| `{k}` | ({gx},{gy},{gz}) | {a[0]} | {a[1] / 1e6:.2f} | {a[1] / a[0] / 1e3:.1f} |")
  txt = "\n".join(lines) + "\n"
  if out:
    open(out, "a").write(txt)
  print(txt)


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "--per-shape":
  per_shape(sys.argv[1], sys.argv[2])
