#!/usr/bin/env python3
"""GPU occupancy of a multi-slot run from a rocprofv3 --kernel-trace database.

usage: tools/rocprof_timeline.py <results.db> [skip-fraction [until-fraction]]
Prints, over the steady-state part of the run: the fraction of wall time with 0, 1, 2, ... kernels resident, the
per-kernel average duration when alone on the device vs overlapped, and the gaps between consecutive kernels of
one stream (dispatch latency the stream pays between dependent kernels).
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = db.execute("select name, stream_id, queue_id, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][3], max(r[4] for r in rows)
until = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lo, hi = t0 + (t1 - t0) * skip, t0 + (t1 - t0) * until
rows = [r for r in rows if r[3] >= lo and r[4] <= hi]
t0, t1 = rows[0][3], max(r[4] for r in rows)
ev = []
for n, s, q, a, b in rows:
    ev.append((a, 1))
    ev.append((b, -1))
ev.sort()
lvl, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[lvl] += t - last
    last = t
    lvl += d
wall = t1 - t0
print(f"window {wall/1e6:.2f} ms, {len(rows)} kernels, streams {len(set(r[1] for r in rows))}, queues {len(set(r[2] for r in rows))}")
for k in sorted(hist):
    print(f"  {k} kernels resident: {100.0*hist[k]/wall:6.2f} %")
bys = collections.defaultdict(list)
for r in rows:
    bys[r[1]].append(r)
gaps = []
for s, lst in bys.items():
    lst.sort(key=lambda r: r[3])
    for a, b in zip(lst, lst[1:]):
        gaps.append((b[3] - a[4], a[0], b[0]))
g = sorted(x[0] for x in gaps)
print(f"gaps between consecutive kernels of a stream: median {g[len(g)//2]/1e3:.2f} us, mean {sum(g)/len(g)/1e3:.2f} us, "
      f"p90 {g[int(len(g)*0.9)]/1e3:.2f} us, sum/stream/wall {sum(g)/len(bys)/wall:.3f}")
dur = collections.defaultdict(list)
for n, s, q, a, b in rows:
    dur[n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]].append(b - a)
tot = sum(sum(v) for v in dur.values())
print(f"sum of kernel durations / wall = {tot/wall:.3f}")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f"  {n[:48]:48s} n={len(v):5d} avg {sum(v)/len(v)/1e3:8.2f} us  share {100.0*sum(v)/tot:5.1f} %")
