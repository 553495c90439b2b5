"""Run ON the GPU box: what one more launch of a kernel costs the PAIR RATE (six pairs in flight), next to the kernel's
duration alone.  Key 15 of ebvo_debug_set launches an idempotent kernel of the chain twice (bit 0 centre, 1 mags, 2 right bank,
3 NCC tile); one context per setting, interleaved and repeated; the counts must not move.
  marginal us per pair = 1e6 / rate(with the repeat) - 1e6 / rate(default)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_based_visual_odometry_amd import _lib, synth  # noqa: E402
from edge_based_visual_odometry_amd.api import Context  # noqa: E402

H, W = synth.SHAPES["kitti"]
F = synth.fundamental_for("kitti")
NS = 6
left, right = synth.stereo_pair("s2", H, W, scene=7, noise_base=0, disparity=12)
NAMES = {0: "default", 1: "centre x 2", 2: "mags x 2", 4: "right bank x 2", 8: "NCC tile x 2", 15: "all four x 2"}
settings = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 15]


def make(mask):
    ctx = Context(H, W, toed_mode="hybrid")
    ctx.set_slots(NS)
    if mask:
        ctx.debug_set(15, mask)
    for k in range(NS):
        ctx.stereo_upload(left, right, slot=k)
    p = ctx.default_params(F)
    p.reserved = _lib.PAIR_NO_SIMS
    return ctx, p


def resident(ctx, p, n, warm):
    sub = done = 0
    t0 = time.perf_counter()
    while done < n + warm:
        while sub < n + warm and sub - done < NS:
            ctx.stereo_submit(p, slot=sub % NS)
            sub += 1
        c = ctx.stereo_wait(slot=done % NS)
        done += 1
        if done == warm:
            t0 = time.perf_counter()
    return n / (time.perf_counter() - t0), c


ctxs = [make(m) for m in settings]
resident(*ctxs[0], 600, 0)  # clocks up
rows = [[] for _ in settings]
counts = []
for rep in range(4):
    for i, (ctx, p) in enumerate(ctxs):
        r, c = resident(ctx, p, 300, 30)
        rows[i].append(r)
        if rep == 0:
            counts.append((c.n_left, c.n_right, c.n_pairs, c.n_matches))
assert all(c == counts[0] for c in counts), counts
med = [sorted(r)[len(r) // 2] for r in rows]
base = med[settings.index(0)] if 0 in settings else None
print("counts", counts[0])
for m, r, v in zip(settings, rows, med):
    extra = "" if base is None or m == 0 else "   marginal %+.1f us per pair" % (1e6 / v - 1e6 / base)
    print("%-16s pairs/s %s   median %.1f%s" % (NAMES.get(m, "mask %d" % m), " ".join("%.0f" % x for x in r), v, extra))
