"""Do the prefetcher's touch kernels run BESIDE the forward's kernels or in line with them?  From a rocprofv3 kernel trace (rocpd SQLite):
  python tools/overlap_check.py gpurun_out/<tag>/prof/bench_results.db
For every touch_ranges_kernel dispatch: its duration and how much of it is covered by other kernels' [start, end) intervals; plus the
idle time of the device between the non-touch kernels around it."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
main = [(s, e, n) for n, s, e in rows if "touch_ranges" not in n]
touch = [(s, e) for n, s, e in rows if "touch_ranges" in n]
print(f"{len(main)} kernels, {len(touch)} touch kernels")
if not touch:
  sys.exit(0)
import bisect
starts = [s for s, _, _ in main]
tot_d = tot_cov = 0
gap_hist = []
for ts, te in touch[len(touch) // 2: len(touch) // 2 + 400]:
  i = bisect.bisect_left(starts, ts)
  cov = 0
  j = max(0, i - 3)
  while j < len(main) and main[j][0] < te:
    s, e, _ = main[j]
    cov += max(0, min(e, te) - max(s, ts))
    j += 1
  tot_d += te - ts
  tot_cov += cov
print(f"touch kernels (sample of 400): mean duration {tot_d / 400 / 1e3:.1f} us, of which covered by other kernels {tot_cov / 400 / 1e3:.1f} us")
# idle gaps of the main chain in the same window
lo, hi = touch[len(touch) // 2][0], touch[min(len(touch) - 1, len(touch) // 2 + 400)][1]
win = [(s, e) for s, e, _ in main if s >= lo and e <= hi]
busy = sum(e - s for s, e in win)
print(f"window: {(hi - lo) / 1e6:.2f} ms, main kernels busy {busy / 1e6:.2f} ms, {len(win)} kernels, mean gap {(hi - lo - busy) / max(1, len(win)) / 1e3:.2f} us")
