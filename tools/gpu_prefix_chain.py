"""Run ON the GPU box: where the pair's time goes IN THE STEADY STATE (six pairs in flight).  A kernel trace gives every
kernel's duration alone; with other pairs' kernels beside it a kernel costs something else (profiles/r04_marginal_cost.txt).
Key 16 of ebvo_debug_set ends the resident pair's chain after stage N, every prefix is a chain the device can run on its own
(the pair's record keeps the counts of the last whole run), and the difference of the pair times of two consecutive prefixes is
what that stage costs.  One context per prefix, interleaved and repeated."""
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
STAGES = [(1, "clear + screen"), (2, "compact_phase"), (3, "exact centre"), (4, "need count / compact"),
          (5, "exact mags"), (6, "exact decide"), (7, "cand_scatter"), (8, "match_prep"), (9, "candidates<count>"),
          (10, "row scan"), (11, "candidates<fill>"), (12, "right bank"), (13, "NCC tile"), (0, "pair_result (whole chain)")]


def make(stop):
    ctx = Context(H, W, toed_mode="hybrid")
    ctx.set_slots(NS)
    for k in range(NS):
        ctx.stereo_upload(left, right, slot=k)
    p = ctx.default_params(F)
    p.reserved = _lib.PAIR_NO_SIMS
    for k in range(NS):           # one whole run per slot: every buffer and the pair's record are in their steady state
        ctx.stereo_submit(p, slot=k)
    for k in range(NS):
        ctx.stereo_wait(slot=k)
    if stop:
        ctx.debug_set(16, stop)
    return ctx, p


def resident(ctx, p, n, warm):
    sub = done = 0
    t0 = time.perf_counter()
    while done < n + warm:
        while sub < n + warm and sub - done < NS:
            ctx.stereo_submit(p, slot=sub % NS)
            sub += 1
        ctx.stereo_wait(slot=done % NS)
        done += 1
        if done == warm:
            t0 = time.perf_counter()
    return (time.perf_counter() - t0) / n * 1e6


ctxs = [make(stop) for stop, _ in STAGES]
resident(*ctxs[-1], 600, 0)  # clocks up
rows = [[] for _ in STAGES]
for rep in range(3):
    for i, (ctx, p) in enumerate(ctxs):
        rows[i].append(resident(ctx, p, 300, 30))
prev = 0.0
print("%-36s %10s %10s" % ("chain ends after", "us / pair", "stage us"))
for (stop, name), r in zip(STAGES, rows):
    v = sorted(r)[len(r) // 2]
    print("%-36s %10.1f %10.1f    (%s)" % (name, v, v - prev, " ".join("%.1f" % x for x in r)))
    prev = v
