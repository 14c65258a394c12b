"""Per-position timeline of ONE UNet forward of the denoise loop from a rocprofv3 kernel trace (rocpd SQLite):
  python tools/forward_timeline.py gpurun_out/prof/bench_results.db [skip_first_forwards]
Splits the trace at sd_stage_kernel / plms_step_kernel (one loop step = stage kernel + UNet forward + PLMS kernel), keeps the
forwards of the most common length (the captured-graph replays), and prints for every launch position its kernel, grid and
the duration averaged over those forwards, with a running sum — the table the per-level / per-block accounting is read from.  Last line: the
gaps between consecutive launches (next start - previous end, under the tracer) and the wall time of a forward (first start to last end)."""
import collections
import re
import sqlite3
import sys


def short(n):
  return re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "")).replace("void ", "")[:60]


def main():
  db = sqlite3.connect(sys.argv[1])
  skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
  rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
  fw, cur = [], None
  for n, s, e, gx, gy, gz, wx in rows:
    sn = short(n)
    if sn.startswith("sd_stage_kernel"):
      cur = []
      fw.append(cur)
    if cur is not None:
      cur.append((sn, e - s, gx // max(wx, 1), gy, gz, s, e))
      if sn.startswith("plms_step_kernel"):
        cur = None
  lens = collections.Counter(len(f) for f in fw)
  L = lens.most_common(1)[0][0]
  good = [f for f in fw if len(f) == L][skip:]
  print(f"# {len(fw)} loop steps in the trace, {len(good)} used ({L} launches each)")
  tot = 0.0
  for i in range(L):
    names = set(f[i][0] for f in good)
    assert len(names) == 1, names
    d = sum(f[i][1] for f in good) / len(good) / 1e3
    tot += d
    print(f"{i:4d} {good[0][i][0]:45s} grid=({good[0][i][2]},{good[0][i][3]},{good[0][i][4]}) {d:8.1f} us  cum {tot / 1e3:7.3f} ms")
  gaps = [f[i + 1][5] - f[i][6] for f in good for i in range(L - 1)]
  wall = sum(f[-1][6] - f[0][5] for f in good) / len(good) / 1e6
  gaps.sort()
  print(f"# gaps between consecutive launches (under the tracer): mean {sum(gaps) / len(gaps) / 1e3:.2f} us, median {gaps[len(gaps) // 2] / 1e3:.2f} us, "
        f"p90 {gaps[len(gaps) * 9 // 10] / 1e3:.2f} us; forward first start -> last end {wall:.3f} ms = kernels {tot / 1e3:.3f} ms + gaps {wall - tot / 1e3:.3f} ms")


if __name__ == "__main__":
  main()
