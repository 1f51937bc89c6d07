#!/usr/bin/env python3
"""What the GPU's queues did during the timed steps of a bench.py run under `rocprofv3 --kernel-trace`:
per queue the kernels, their busy time and the gaps between them; over all queues the time with n kernels running.

    python tools/timeline.py kernel_trace.csv [dump]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void ", "")[:28]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short))
ev.sort()
# the middle half of the sketch kernels' time span = steady state
sk = [e for e in ev if e[3].startswith("sketch")]
# bench.py: 5 warm-up steps, then the timed steps (pipelined), then as many blocking ones (at most 200) for the
# per-kernel table: the timed ones are the first half of what follows the warm-up
n_timed = (len(sk) - 5) // 2
a, b = 5 + n_timed // 4, 5 + 3 * n_timed // 4
t0, t1 = sk[a][0], sk[b][0]
n_steps = b - a
span = t1 - t0
print("steady window: %.3f ms, %d sketch launches -> %.4f ms per step" % (span / 1e6, n_steps, span / 1e6 / n_steps))
perq = defaultdict(list)
for s, e, q, n in ev:
    if e > t0 and s < t1:
        perq[q].append((s, e, n))
for q, lst in sorted(perq.items()):
    busy = sum(min(e, t1) - max(s, t0) for s, e, n in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    byname = defaultdict(lambda: [0, 0])
    for s, e, n in lst:
        byname[n][0] += 1
        byname[n][1] += e - s
    print("queue %s: %d kernels, busy %.2f of the window, median gap %.1f us, mean gap %.1f us" % (
        q, len(lst), busy / span, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, sum(gaps) / max(len(gaps), 1) / 1e3))
    for n, (c, t) in sorted(byname.items(), key=lambda x: -x[1][1]):
        print("      %-28s %4d x %8.1f us" % (n, c, t / c / 1e3))
# concurrency histogram
pts = []
for s, e, q, n in ev:
    if e > t0 and s < t1:
        pts.append((max(s, t0), 1))
        pts.append((min(e, t1), -1))
pts.sort()
hist = defaultdict(int)
cur, last = 0, t0
for t, d in pts:
    hist[cur] += t - last
    last = t
    cur += d
hist[cur] += t1 - last
print("kernels running at once: " + "  ".join("%d: %.2f" % (k, v / span) for k, v in sorted(hist.items())))
if len(sys.argv) > 2:
    mid = (t0 + t1) // 2
    for s, e, q, n in ev:
        if s > mid and s < mid + 3 * span // n_steps:
            print("%10.1f %10.1f  q%s %s" % ((s - mid) / 1e3, (e - mid) / 1e3, q, n))
