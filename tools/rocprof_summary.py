#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (gpurun_out/prof/*) into tracked text files under profiles/.

usage: tools/rocprof_summary.py <round-tag>   e.g. r01
Reads  gpurun_out/prof/trace/*_results.db      (rocprofv3 --kernel-trace --stats)
       gpurun_out/prof/pmc_fetch/*_results.db  (rocprofv3 --kernel-trace --pmc FETCH_SIZE)
       gpurun_out/prof/pmc_write/*_results.db  (rocprofv3 --kernel-trace --pmc WRITE_SIZE)
Writes profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc_hbm.txt and profiles/conv_pmc.json
(the HBM traffic per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950:
FETCH_SIZE x 2, WRITE_SIZE as read; both are reported by rocprofv3 in KiB).
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
mode = sys.argv[2] if len(sys.argv) > 2 else "hybrid"          # toed mode of the profiled run
dominant = sys.argv[3] if len(sys.argv) > 3 else "toed_exact_centre"   # bench.py's name of the dominant kernel
symbol = sys.argv[4] if len(sys.argv) > 4 else dominant + "_kernel"   # its HIP symbol (substring)
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def db(name):
    f = glob.glob(os.path.join(src, name, "*_results.db"))
    return sqlite3.connect(f[0]) if f else None


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


out = []
t = db("trace")
if t:
    out.append(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --toed-mode {mode}   [{tag}]")
    out.append(f"{'kernel':40s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
    for name, calls, total, avg, pct in t.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out.append(f"{short(name):40s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
    open(os.path.join(dst, f"{tag}_kernel_stats.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))

pm = []
res = {}
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    d = db(name)
    if not d:
        continue
    pm.append(f"# rocprofv3 --kernel-trace --pmc {counter} -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --toed-mode {mode}   [{tag}]")
    pm.append(f"{'kernel':40s} {'launches':>8s} {'avg KiB/launch':>16s}")
    rows = d.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                     "group by kernel_name order by avg(value) desc", (counter,)).fetchall()
    for k, n, v in rows:
        pm.append(f"{short(k):40s} {n:8d} {v:16.1f}")
        if symbol in k:
            res[counter] = v * 1024.0
if pm:
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        fetch, write = res["FETCH_SIZE"] * 2.0, res["WRITE_SIZE"]
        pm.append("")
        pm.append(f"{dominant} per launch (2 images): FETCH_SIZE {res['FETCH_SIZE']/1e6:.2f} MB raw -> x2 (gfx950 correction) "
                  f"= {fetch/1e6:.2f} MB;  WRITE_SIZE {write/1e6:.2f} MB;  HBM traffic = {(fetch+write)/1e6:.2f} MB")
        json.dump({"kernel": dominant, "toed_mode": mode,
                   "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {tag}; FETCH_SIZE doubled per MI355X_MICROARCH.md",
                   "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                   "hbm_bytes_per_launch": fetch + write}, open(os.path.join(dst, "dominant_pmc.json" if mode == "hybrid" else f"dominant_pmc_{mode}.json"), "w"), indent=1)
    open(os.path.join(dst, f"{tag}_pmc_hbm.txt"), "w").write("\n".join(pm) + "\n")
    print("\n".join(pm))
