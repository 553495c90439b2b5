"""Where and when the waves of the two exact TOED kernels ran (developer tool; needs a library built with
`make -C edge_based_visual_odometry_amd/csrc clean all EXTRA=-DEBVO_WAVE_TRACE`).  Run ON the GPU box."""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from edge_based_visual_odometry_amd import api, synth  # noqa: E402

h, w = synth.SHAPES["kitti"]
FAKE = len(sys.argv) > 1 and sys.argv[1] == "--fake"
ctx = None if FAKE else api.Context(h, w, toed_mode="hybrid")
l, r = synth.stereo_pair("s2", h, w, scene=7, noise_base=0, disparity=12)
F = synth.fundamental_for("kitti")
if FAKE:
    def fn(which, ptr, n):
        rng = np.random.default_rng(which)
        a = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(n, 4))
        a[:, 0] = 10 ** 9 + rng.integers(0, 300, n)
        a[:, 1] = a[:, 0] + rng.integers(3000, 9000, n)
        a[:, 2] = (rng.integers(0, 8, n).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 15, n).astype(np.uint64)
        a[:, 3] = rng.integers(0, 3, n)
        return 0
else:
    ctx.stereo_upload(l, r)
    for _ in range(3):
        ctx.stereo_run(ctx.default_params(F))
    fn = ctx.lib.ebvo_wave_trace_read
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
for which, name in ((0, "exact_centre"), (1, "exact_mags")):
    buf = np.zeros((8192, 4), dtype=np.uint64)
    rc = fn(which, buf.ctypes.data, 8192)
    assert rc == 0, rc
    buf = buf[buf[:, 1] > 0]
    t0, t1, hw, tasks = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), buf[:, 2], buf[:, 3].astype(np.int64)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    base = t0.min()  # s_memrealtime: one 100 MHz counter for the whole device (10 ns ticks)
    dur = t1 - t0
    work = tasks > 0
    print(f"== {name}: {len(buf)} waves recorded, {work.sum()} with work, tasks {tasks.sum()}; kernel span {(t1 - base).max() / 100:.1f} us")
    print("   start offset of working waves: min/median/p90/max", np.percentile(t0[work] - base, [0, 50, 90, 100]) / 100, "us")
    print("   duration of working waves:     min/median/p90/max", np.percentile(dur[work], [0, 10, 50, 90, 100]) / 100, "us")
    print("   end offset of working waves:   min/median/p90/max", np.percentile(t1[work] - base, [0, 10, 50, 90, 100]) / 100, "us")
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = {}
    for k, s_, tk in zip(key[work], simd[work], tasks[work]):
        per_cu.setdefault(int(k), [0, 0, 0, 0])[int(s_)] += int(tk)
    loads = np.array(list(per_cu.values()))
    print(f"   CUs seen {len(per_cu)}; tasks per CU min/median/max {loads.sum(1).min()} {int(np.median(loads.sum(1)))} {loads.sum(1).max()};"
          f" tasks per SIMD min/median/max {loads.min()} {int(np.median(loads))} {loads.max()}")
    hist = np.bincount(loads.ravel())
    print("   histogram of tasks per SIMD:", dict(enumerate(hist.tolist())))
    print("   xcc ids", sorted(set(xcc.tolist())), "se", sorted(set(se.tolist())), "sh", sorted(set(sh.tolist())), "cu", sorted(set(cu.tolist())))
