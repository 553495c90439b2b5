"""Run ON the GPU box: the resident pair rate (six pairs in flight) under settings of the developer keys of ebvo_debug_set,
one context per setting, interleaved and repeated (box-to-box drift is 2-3 %, minute-to-minute ~1 %); the counts must not move.
usage: python3 tools/gpu_ab_keys.py default 11=768 11=768,12=1280 17=4096 ...      (key=value[,key=value...]; `default` = no key)
  11 / 12  grid of toed_exact_centre / toed_exact_mags in blocks      17  grid of ncc_tile_kernel in blocks
  14       1 = lines, boxes, sincos, row pairs as four launches       15  bit mask: an idempotent kernel launched twice"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_based_visual_odometry_amd import _lib, synth  # noqa: E402
from edge_based_visual_odometry_amd.api import Context  # noqa: E402

H, W = synth.SHAPES["kitti"]
F = synth.fundamental_for("kitti")
NS = int(os.environ.get("EBVO_AB_SLOTS", "6"))   # pairs in flight
left, right = synth.stereo_pair("s2", H, W, scene=7, noise_base=0, disparity=12)
settings = sys.argv[1:] or ["default"]


def make(spec):
    ctx = Context(H, W, toed_mode="hybrid")
    ctx.set_slots(NS)
    if spec != "default":
        for kv in spec.split(","):
            k, v = kv.split("=")
            ctx.debug_set(int(k), int(v))
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


ctxs = [make(sp) for sp in settings]
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
print("lib:", os.environ.get("EBVO_LIB", "(tree)"), " counts", counts[0])
for sp, r in zip(settings, rows):
    v = sorted(r)[len(r) // 2]
    print("%-24s pairs/s %s   median %.1f   %.1f us / pair" % (sp, " ".join("%.0f" % x for x in r), v, 1e6 / v))
