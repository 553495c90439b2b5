"""Run ON the GPU box: what a ctypes call into the library costs in a process where PyTorch has initialised the device."""
import ctypes as C
import sys
import threading
import time

import numpy as np

if "torch" in sys.argv:
    import torch
    torch.cuda.synchronize()
sys.path.insert(0, ".")
from edge_based_visual_odometry_amd import synth  # noqa: E402
from edge_based_visual_odometry_amd.api import Context, ptr  # noqa: E402

H, W = synth.SHAPES["kitti"]
ctx = Context(H, W, toed_mode="hybrid")
ctx.set_slots(2)
l, r = (np.ascontiguousarray(im) for im in synth.stereo_pair("s2", H, W))
print("python threads:", [t.name for t in threading.enumerate()], "switch interval", sys.getswitchinterval(), flush=True)


def t_of(fn, n=300):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


lib, cp = ctx.lib, ctx._ctx
pl, pr = ptr(l), ptr(r)
for label in ("pageable (not registered)", "registered"):
    if label == "registered":
        ctx.host_register(l)
        ctx.host_register(r)
    print(label)
    print("   trivial C call (ebvo_get_toed_mode)        %7.1f us" % t_of(lambda: lib.ebvo_get_toed_mode(cp)))
    print("   raw ebvo_stereo_upload_async, cached args  %7.1f us" % t_of(lambda: lib.ebvo_stereo_upload_async(cp, 1, pl, pr, H, W, W, W)))
    print("   wrapper ctx.stereo_upload_async            %7.1f us" % t_of(lambda: ctx.stereo_upload_async(l, r, slot=1)))
    print("   ptr(l); ptr(r)                             %7.1f us" % t_of(lambda: (ptr(l), ptr(r))))
    print("   l.flags.c_contiguous, l.dtype == uint8     %7.1f us" % t_of(lambda: (l.flags.c_contiguous, l.dtype == np.uint8)))
    print("   np.zeros(4)                                %7.1f us" % t_of(lambda: np.zeros(4)))
ctx.host_unregister(l)
ctx.host_unregister(r)
ctx.stereo_upload(l, r, slot=1)
ctx.close()
