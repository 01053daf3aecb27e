"""GPU timeline of a multi-context bench run from a rocprofv3 kernel trace: how much of the wall time some kernel was running, how many
kernels overlapped, and which gaps were left.  Usage: python scripts/timeline.py <kernel_trace.csv> [n_steps_to_analyse]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows]
ev.sort()
# the timed region: from the last third of seed_kernel launches on
seeds = [e for e in ev if e[2].endswith("seed_kernel")]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else max(3, len(seeds) // 3)
t0 = seeds[-n_last][0]
t1 = max(e[1] for e in ev)
win = [e for e in ev if e[0] >= t0]
tot = t1 - t0
# union and concurrency histogram
pts = []
for s, e, _, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
conc = defaultdict(int)
cur, last = 0, t0
for t, d in pts:
    conc[cur] += t - last
    cur += d; last = t
print("window %.1f ms, %d kernels, %d seed_kernel launches" % (tot / 1e6, len(win), n_last))
for k in sorted(conc):
    print("  %d kernels running: %6.2f %% of the window" % (k, 100.0 * conc[k] / tot))
busy = tot - conc[0]
ksum = sum(e - s for s, e, _, _ in win)
print("GPU busy (some kernel running) %.1f ms = %.1f %%; sum of kernel durations %.1f ms (%.2fx the busy time)" % (busy / 1e6, 100.0 * busy / tot, ksum / 1e6, ksum / max(1, busy)))
per = defaultdict(lambda: [0, 0])
for s, e, n, _ in win:
    per[n][0] += e - s; per[n][1] += 1
print("per kernel (ms total in window, launches, avg us):")
for n, (d, c) in sorted(per.items(), key=lambda x: -x[1][0])[:22]:
    print("  %-34s %8.2f %6d %9.1f" % (n[-34:], d / 1e6, c, d / c / 1e3))
# idle gaps
gaps = []
cur, last = 0, t0
for t, d in pts:
    if cur == 0 and t > last:
        gaps.append(t - last)
    cur += d; last = t
gaps.sort(reverse=True)
print("idle gaps: %d, total %.2f ms; largest (us): %s" % (len(gaps), sum(gaps) / 1e6, [round(g / 1e3) for g in gaps[:12]]))
