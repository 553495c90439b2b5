"""A/B of round 4's switches on ONE box (run ON the GPU box): the resident pair loop and the streamed-ingest loop with
  key 14 = 1  lines, boxes, sincos and row pairs as four launches instead of one (match_prep_kernel),
  key 13 = 1  ebvo_stereo_upload_async through the upload stream instead of the pull kernel,
  NO_SIMS off the four similarities stored (round 3),
each against the defaults, interleaved and repeated (box-to-box and minute-to-minute drift is ~1 %)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from edge_based_visual_odometry_amd import _lib, synth  # noqa: E402
from edge_based_visual_odometry_amd.api import Context  # noqa: E402

sys.path.insert(0, ".")
import bench  # noqa: E402

H, W = synth.SHAPES["kitti"]
F = synth.fundamental_for("kitti")
NS = 6
ctx = Context(H, W, toed_mode="hybrid")
ctx.set_slots(NS + 1)
left, right = synth.stereo_pair("s2", H, W, scene=7, noise_base=0, disparity=12)
ring = [tuple(np.ascontiguousarray(im) for im in synth.stereo_pair("s2", H, W, scene=7, noise_base=10 * k, disparity=12)) for k in range(4)]
for pair in ring:
    for im in pair:
        ctx.host_register(im)
p_full = ctx.default_params(F)
p_nosims = ctx.default_params(F)
p_nosims.reserved = _lib.PAIR_NO_SIMS


def resident(params, n):
    for k in range(NS):
        ctx.stereo_upload(left, right, slot=k)
    sub = done = 0
    t0 = None
    warm = 5 * NS
    while done < n + warm:
        while sub < n + warm and sub - done < NS:
            ctx.stereo_submit(params, slot=sub % NS)
            sub += 1
        ctx.stereo_wait(slot=done % NS)
        done += 1
        if done == warm:
            t0 = time.perf_counter()
    return n / (time.perf_counter() - t0)


def ingest(params, n, fetch=None):
    bench.ingest_loop(ctx, params, ring, NS, 4 * (NS + 1), fetch)
    t, _ = bench.ingest_loop(ctx, params, ring, NS, n, fetch)
    return n / t


resident(p_nosims, 600)                      # clocks up
configs = [("default", None, p_nosims), ("four prep launches (14)", 14, p_nosims),
           ("sims stored", None, p_full)]
rows = {name: [] for name, _, _ in configs}
for rep in range(4):
    for name, key, prm in configs:
        if key:
            ctx.debug_set(key, 1)
        rows[name].append(resident(prm, 300))
        if key:
            ctx.debug_set(key, 0)
print("resident loop, 300 pairs, pairs/s (4 interleaved repetitions)")
for name, v in rows.items():
    print(f"  {name:20s} {np.median(v):8.1f}   " + " ".join(f"{x:7.1f}" for x in v))
ing = {"pull (default)": [], "upload stream (13)": [], "pull + compact fetch": [], "stream + compact fetch": [], "sync upload (frame_loop)": []}
pool = [tuple(np.array(im) for im in pair) for pair in ring]
for rep in range(3):
    ing["pull (default)"].append(ingest(p_nosims, 200))
    ing["pull + compact fetch"].append(ingest(p_nosims, 200, "compact"))
    ctx.debug_set(13, 1)
    ing["upload stream (13)"].append(ingest(p_nosims, 200))
    ing["stream + compact fetch"].append(ingest(p_nosims, 200, "compact"))
    ctx.debug_set(13, 0)
    t, _ = bench.frame_loop(ctx, p_nosims, pool, NS, 200, True, None)
    ing["sync upload (frame_loop)"].append(200 / t)
print("streamed ingest, 200 pairs, pairs/s")
for name, v in ing.items():
    print(f"  {name:26s} {np.median(v):8.1f}   " + " ".join(f"{x:7.1f}" for x in v))
for k in range(NS + 1):
    ctx.stereo_upload(left, right, slot=k)
for pair in ring:
    for im in pair:
        ctx.host_unregister(im)
ctx.close()
