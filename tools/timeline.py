#!/usr/bin/env python3
"""Tuning helper: kernel timeline of the pipelined bench from a rocprofv3 --kernel-trace database — for a window in the steady state,
every dispatch with its start (us since the window began), duration and queue, plus how many kernels ran at each moment.
usage: tools/timeline.py <results.db> [first_step] [n_steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first, nst = (int(sys.argv[2]) if len(sys.argv) > 2 else 40), (int(sys.argv[3]) if len(sys.argv) > 3 else 4)
rows = [(n.replace("void ", "").replace("wsa::", "").split("(")[0][:34], st, en, q, sid) for n, st, en, q, sid in
        db.execute("select name, start, end, queue_id, stream_id from kernels order by start") if "wsa::" in n or (len(sys.argv) > 4 and sys.argv[4] == "all")]
fe = [i for i, r in enumerate(rows) if r[0].startswith("fe_kernel")]
i0, i1 = fe[first], fe[first + nst]
t0 = rows[i0][1]
print("start_us  dur_us  queue  kernel")
for r in rows[i0:i1]:
    print("%8.1f %7.1f  %5s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0]))
span = (rows[i1][1] - t0) / 1e3
print("window %.1f us for %d steps = %.1f us per step" % (span, nst, span / nst))
# concurrency histogram
ev = []
for r in rows[i0:i1]:
    ev.append((r[1], 1)); ev.append((r[2], -1))
ev.sort()
cur, last, hist = 0, t0, {}
for t, d in ev:
    if t > rows[i1][1]:
        break
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
tot = sum(hist.values())
print("kernels running at once:", {k: "%.0f%%" % (100 * v / tot) for k, v in sorted(hist.items())})
