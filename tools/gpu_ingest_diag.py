"""Run ON the GPU box: the streamed-ingest loops in bench.py's order, with the share of pairs that went out as a captured graph."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (a profiler may start the tool from /tmp)
from edge_based_visual_odometry_amd import _lib, synth  # noqa: E402
from edge_based_visual_odometry_amd.api import Context  # noqa: E402
import bench  # noqa: E402

if "torch" in sys.argv:                        # bench.py's process: torch has initialised the device before the library does
    import torch
    torch.cuda.synchronize()
    print("torch initialised the device first", flush=True)
H, W = synth.SHAPES["kitti"]
F = synth.fundamental_for("kitti")
NS = 6
ctx = Context(H, W, toed_mode="hybrid")
ctx.set_slots(NS)
left, right = synth.stereo_pair("s2", H, W, scene=7, noise_base=0, disparity=12)
for k in range(NS):
    ctx.stereo_upload(left, right, slot=k)
p = ctx.default_params(F)
p.reserved = _lib.PAIR_NO_SIMS


def resident(n):
    sub = done = 0
    t0 = time.perf_counter()
    while done < n:
        while sub < n and sub - done < NS:
            ctx.stereo_submit(p, slot=sub % NS)
            sub += 1
        ctx.stereo_wait(slot=done % NS)
        done += 1
    return n / (time.perf_counter() - t0)


def show(name, fn):
    g0 = ctx.graph_launches
    t0 = time.perf_counter()
    r = fn()
    print(f"{name:46s} {r if isinstance(r, str) else f'{r:8.1f} pairs/s'}   graphs +{ctx.graph_launches - g0}   {time.perf_counter() - t0:.3f} s", flush=True)


show("resident 600 (warm)", lambda: resident(600))
show("resident 300", lambda: resident(300))
ring = [tuple(np.ascontiguousarray(im) for im in synth.stereo_pair("s2", H, W, scene=7, noise_base=10 * k, disparity=12)) for k in range(4)]
for pair in ring:
    for im in pair:
        ctx.host_register(im)
ctx.set_slots(NS + 1)
flags = ("push" if "push" in sys.argv else "compact")
PACK = "pack" in sys.argv
pf = ctx.default_params(F)                    # the fetch loops: with EBVO_PAIR_PUSH when the results are pushed
pf.reserved = p.reserved | (_lib.PAIR_PUSH if flags == "push" else (_lib.PAIR_PACK if PACK else 0))


def ing(n, fetch=None):
    t, _ = bench.ingest_loop(ctx, pf if fetch else p, ring, NS, n, fetch)
    return n / t


t0 = time.perf_counter()
for _ in range(200):
    ctx.stereo_upload_async(*ring[0], slot=NS)
print(f"host time of ebvo_stereo_upload_async: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call; by form: {ctx.ingest_stats()}", flush=True)
show("ingest warm 28", lambda: ing(28))
show("ingest 20", lambda: ing(20))
hu = {}
show("ingest 300", lambda: 300 / bench.ingest_loop(ctx, p, ring, NS, 300, None, hu)[0])
print("   host us per pair:", {k: round(v, 1) for k, v in hu.items()}, flush=True)
if "torchlate" in sys.argv:                    # bench.py's order: the library's context exists before torch touches the device
    import torch
    torch.cuda.synchronize()
    print("torch initialised the device now", flush=True)
    hu = {}
    show("ingest 300 (torch now active)", lambda: 300 / bench.ingest_loop(ctx, p, ring, NS, 300, None, hu)[0])
    print("   host us per pair:", {k: round(v, 1) for k, v in hu.items()}, flush=True)
show(f"ingest + {flags} warm 7", lambda: ing(7, flags))
show(f"ingest + {flags} 60", lambda: ing(60, flags))
show(f"ingest + {flags} 300", lambda: ing(300, flags))
show("ingest 300 (after the fetch loops)", lambda: ing(300))
show("ingest 300 again", lambda: ing(300))
print("uploads by form:", ctx.ingest_stats(), flush=True)
show("resident 300 (7 slots exist now)", lambda: resident(300))
for k in range(NS + 1):
    ctx.stereo_upload(left, right, slot=k)
for pair in ring:
    for im in pair:
        ctx.host_unregister(im)
ctx.close()
