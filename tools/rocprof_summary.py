#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (gpurun_out/prof/*) into tracked text files under profiles/.

usage: tools/rocprof_summary.py <round-tag> <toed-mode> [streams]
Reads  gpurun_out/prof/trace_<mode>/*_results.db      (rocprofv3 --kernel-trace --stats)
       gpurun_out/prof/pmc_fetch_<mode>/*_results.db  (rocprofv3 --kernel-trace --pmc FETCH_SIZE)
       gpurun_out/prof/pmc_write_<mode>/*_results.db  (rocprofv3 --kernel-trace --pmc WRITE_SIZE)
Writes profiles/<tag>_kernel_stats_<mode>.txt, profiles/<tag>_pmc_hbm_<mode>.txt and profiles/kernel_pmc_<mode>.json
(HBM bytes per launch of every kernel symbol, corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE x 2,
WRITE_SIZE as read; both are reported by rocprofv3 in KiB).  bench.py sums the symbols behind its dominant kernel id.
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
mode = sys.argv[2] if len(sys.argv) > 2 else "hybrid"
streams = sys.argv[3] if len(sys.argv) > 3 else "1"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def db(name):
    f = glob.glob(os.path.join(src, name, "*_results.db"))
    return sqlite3.connect(f[0]) if f else None


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


t = db(f"trace_{mode}")
if t:
    out = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 60 --warmup 3 --no-cpu-baseline "
           f"--toed-mode {mode} --streams {streams}   [{tag}]",
           f"{'kernel':40s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}"]
    for name, calls, total, avg, pct in t.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out.append(f"{short(name):40s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
    open(os.path.join(dst, f"{tag}_kernel_stats_{mode}.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))

t = db("trace_default") if mode == "hybrid" else None
if t:
    out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (the default run: hybrid TOED, 3 pairs "
           f"in flight; kernels of different pairs overlap, and the tracer itself slows the overlap)   [{tag}]",
           f"{'kernel':40s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}"]
    for name, calls, total, avg, pct in t.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out.append(f"{short(name):40s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
    open(os.path.join(dst, f"{tag}_kernel_stats_default.txt"), "w").write("\n".join(out) + "\n")

pm, res = [], {}
for name, counter in ((f"pmc_fetch_{mode}", "FETCH_SIZE"), (f"pmc_write_{mode}", "WRITE_SIZE")):
    d = db(name)
    if not d:
        continue
    pm.append(f"# rocprofv3 --kernel-trace --pmc {counter} -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "
              f"--toed-mode {mode} --streams 1   [{tag}]")
    pm.append(f"{'kernel':40s} {'launches':>8s} {'avg KiB/launch':>16s}")
    rows = d.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                     "group by kernel_name order by avg(value) desc", (counter,)).fetchall()
    for k, n, v in rows:
        pm.append(f"{short(k):40s} {n:8d} {v:16.1f}")
        res.setdefault(short(k), {})[counter] = v * 1024.0
if pm:
    table = {}
    for k, v in res.items():
        f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
        if f is None or w is None:
            continue
        table[k] = {"fetch_bytes_per_launch": f * 2.0, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f * 2.0 + w}
    json.dump({"toed_mode": mode,
               "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, {tag}; FETCH_SIZE doubled "
                         "(gfx950 correction, MI355X_MICROARCH.md); bytes per launch of each kernel symbol",
               "kernels": table}, open(os.path.join(dst, f"kernel_pmc_{mode}.json"), "w"), indent=1)
    tot = sum(v["hbm_bytes_per_launch"] for v in table.values())
    pm.append("")
    pm.append(f"sum over the kernel symbols of one launch each (FETCH x2 + WRITE): {tot / 1e6:.1f} MB")
    open(os.path.join(dst, f"{tag}_pmc_hbm_{mode}.txt"), "w").write("\n".join(pm) + "\n")
    print("\n".join(pm))
