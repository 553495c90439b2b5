#!/usr/bin/env python3
"""Summarise the rocprofv3 rocpd databases written by tools/gpu_profile.sh into tracked text files under profiles/.

usage: tools/rocprof_summary.py <tag> <toed-mode> [suffix]
Reads  gpurun_out/prof_<tag>/trace_<mode>/**/*_results.db       (rocprofv3 --kernel-trace --stats)
       gpurun_out/prof_<tag>/pmc_fetch_<mode>, pmc_write_<mode>   (FETCH_SIZE / WRITE_SIZE, separate passes)
       gpurun_out/prof_<tag>/pmc_sq1.._sq3_<mode>, pmc_grbm_<mode> (SQ / GRBM counters, eight per pass)
Writes profiles/<tag>_kernel_stats_<mode><suffix>.txt, profiles/<tag>_pmc_hbm_<mode><suffix>.txt,
       profiles/<tag>_pmc_sq_<mode><suffix>.txt and (without a suffix) profiles/kernel_pmc_<mode>.json: HBM bytes per launch
       of every kernel symbol, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2, WRITE_SIZE as
       read; both are reported by rocprofv3 in KiB).  bench.py sums the symbols behind its dominant kernel id.
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
mode = sys.argv[2] if len(sys.argv) > 2 else "hybrid"
suffix = sys.argv[3] if len(sys.argv) > 3 else ""
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.environ.get("EBVO_PROFILES_DST", os.path.join(ROOT, "profiles"))  # on the GPU box: a directory under gpurun_out/
os.makedirs(dst, exist_ok=True)


def db(name):
    f = sorted(glob.glob(os.path.join(src, name, "**", "*_results.db"), recursive=True))
    if len(f) > 1:  # a child process of the profiled program wrote its own database: which one is bench.py's is a guess
        sys.exit(f"rocprof_summary: {len(f)} databases under {os.path.join(src, name)}: {f}; profile a run without child "
                 "GPU processes (bench.py --no-transfer-legs)")
    return sqlite3.connect(f[0]) if f else None


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def command_of(name):
    log = os.path.join(src, "log.txt")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith(f"== {name}:"):
                return line.split(":", 1)[1].strip()
    return name


t = db(f"trace_{mode}")
durations = {}
if t:
    out = [f"# {command_of('trace_' + mode)}   [{tag}]",
           f"{'kernel':44s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}"]
    for name, calls, total, avg, pct in t.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out.append(f"{short(name):44s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
        durations[short(name)] = avg
    open(os.path.join(dst, f"{tag}_kernel_stats_{mode}{suffix}.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


def counters(name):
    """{kernel: {counter: (launches, average per launch)}}"""
    d = db(name)
    res = {}
    if not d:
        return res
    rows = d.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                     "group by kernel_name, counter_name").fetchall()
    for k, c, n, v in rows:
        res.setdefault(short(k), {})[c] = (n, v)
    return res


pm, res = [], {}
for name, counter in ((f"pmc_fetch_{mode}", "FETCH_SIZE"), (f"pmc_write_{mode}", "WRITE_SIZE")):
    c = counters(name)
    if not c:
        continue
    pm.append(f"# {command_of(name)}   [{tag}]")
    pm.append(f"{'kernel':44s} {'launches':>8s} {'avg KiB/launch':>16s}")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1].get(counter, (0, 0))[1]):
        if counter in v:
            pm.append(f"{k:44s} {v[counter][0]:8d} {v[counter][1]:16.1f}")
            res.setdefault(k, {})[counter] = v[counter][1] * 1024.0
if pm:
    table = {}
    for k, v in res.items():
        f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
        if f is None or w is None:
            continue
        table[k] = {"fetch_bytes_per_launch": f * 2.0, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f * 2.0 + w}
    tot = sum(v["hbm_bytes_per_launch"] for k, v in table.items() if not k.startswith("__amd"))
    pm.append("")
    pm.append(f"sum over the kernel symbols of one launch each (FETCH x2 + WRITE): {tot / 1e6:.1f} MB")
    open(os.path.join(dst, f"{tag}_pmc_hbm_{mode}{suffix}.txt"), "w").write("\n".join(pm) + "\n")
    print("\n".join(pm))
    if not suffix and "_" not in tag:  # the machine-readable table bench.py reads: the headline workload only
        json.dump({"toed_mode": mode,
                   "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, {tag}; FETCH_SIZE doubled "
                             "(gfx950 correction, MI355X_MICROARCH.md); bytes per launch of each kernel symbol",
                   "kernels": table}, open(os.path.join(dst, f"kernel_pmc_{mode}.json"), "w"), indent=1)

# ---- SQ counters: what the waves of each kernel spent their time on --------------------------------------------------
sq = {}
cmds = []
for name in (f"pmc_sq1_{mode}", f"pmc_sq2_{mode}", f"pmc_sq3_{mode}", f"pmc_grbm_{mode}"):
    c = counters(name)
    if c:
        cmds.append(f"# {command_of(name)}   [{tag}]")
    for k, v in c.items():
        sq.setdefault(k, {}).update({cn: val for cn, (n, val) in v.items()})
if sq:
    lines = cmds + [
        "#",
        "# per launch, summed over the chip.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles",
        "# (MI355X_MICROARCH.md).  valu/wave = SQ_INSTS_VALU / SQ_WAVES; issue% = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves",
        "# sharing a SIMD is not separable here, so the columns below are ratios of the raw sums:",
        "#   act_valu%  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (share of wave lifetime spent issuing VALU)",
        "#   act_any%   = SQ_ACTIVE_INST_ANY  / SQ_WAVE_CYCLES",
        "#   wait_inst% = SQ_WAIT_INST_ANY    / SQ_WAVE_CYCLES   (issue stalls: dependency / pipe busy)",
        "#   wait_any%  = SQ_WAIT_ANY         / SQ_WAVE_CYCLES   (parked on s_waitcnt / barrier)",
        "#   valu_busy% = 4 x SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES x 4 SIMDs / CUs-normalised) is not derivable without the",
        "#                per-SIMD split; use `simd_valu%` = SQ_INST_CYCLES_VALU / (4 x SQ_BUSY_CU_CYCLES-equivalent) only as",
        "#                a trend.  The trace's avg_us is printed next to the counters for the absolute scale.",
        f"{'kernel':34s} {'avg_us':>8s} {'waves':>9s} {'valu/wave':>10s} {'salu/wave':>10s} {'lds/wave':>9s} {'vmem/wave':>10s} "
        f"{'act_valu%':>9s} {'act_any%':>9s} {'wait_inst%':>10s} {'wait_any%':>9s} {'f64 add':>10s} {'f64 mul':>10s} "
        f"{'f64 fma':>10s} {'cvt':>10s} {'int32':>10s} {'lds_conf%':>9s}"]
    def g(d, k):
        return d.get(k, float("nan"))
    for k, d in sorted(sq.items(), key=lambda kv: -durations.get(kv[0], 0.0)):
        if k.startswith("__amd"):
            continue
        wv = g(d, "SQ_WAVES") or float("nan")
        wc = g(d, "SQ_WAVE_CYCLES") or float("nan")
        vmem = g(d, "SQ_INSTS_VMEM_RD") + g(d, "SQ_INSTS_VMEM_WR")
        lds_idx = g(d, "SQ_LDS_IDX_ACTIVE")
        conf = 100.0 * g(d, "SQ_LDS_BANK_CONFLICT") / lds_idx if lds_idx and lds_idx == lds_idx and lds_idx > 0 else 0.0
        lines.append(f"{k[:34]:34s} {durations.get(k, float('nan')):8.2f} {wv:9.0f} {g(d, 'SQ_INSTS_VALU') / wv:10.1f} "
                     f"{g(d, 'SQ_INSTS_SALU') / wv:10.1f} {g(d, 'SQ_INSTS_LDS') / wv:9.1f} {vmem / wv:10.1f} "
                     f"{100 * g(d, 'SQ_ACTIVE_INST_VALU') / wc:9.1f} {100 * g(d, 'SQ_ACTIVE_INST_ANY') / wc:9.1f} "
                     f"{100 * g(d, 'SQ_WAIT_INST_ANY') / wc:10.1f} {100 * g(d, 'SQ_WAIT_ANY') / wc:9.1f} "
                     f"{g(d, 'SQ_INSTS_VALU_ADD_F64'):10.0f} {g(d, 'SQ_INSTS_VALU_MUL_F64'):10.0f} "
                     f"{g(d, 'SQ_INSTS_VALU_FMA_F64'):10.0f} {g(d, 'SQ_INSTS_VALU_CVT'):10.0f} "
                     f"{g(d, 'SQ_INSTS_VALU_INT32'):10.0f} {conf:9.1f}")
    lines.append("")
    lines.append("# raw per-launch averages")
    names = sorted({c for d in sq.values() for c in d})
    for k, d in sorted(sq.items(), key=lambda kv: -durations.get(kv[0], 0.0)):
        if k.startswith("__amd"):
            continue
        lines.append(k + ": " + ", ".join(f"{c}={d[c]:.0f}" for c in names if c in d))
    open(os.path.join(dst, f"{tag}_pmc_sq_{mode}{suffix}.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:len(cmds) + 40]))
