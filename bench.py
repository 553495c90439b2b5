#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the edge-extraction-and-matching hot path on MI355X.

A "step" is one pass of the whole hot path over one synthetic stereo pair that is already resident in HBM:
TOED(left) + TOED(right) + epipolar/disparity/orientation candidate search + NCC scoring (ebvo_stereo_submit /
ebvo_stereo_wait).  One process per GPU, one sequence per GPU, no collective on the data path (torch.distributed is
used only for the barrier and the max over ranks of the timed region).

Prints ONE JSON line on rank 0:
  value / ms_per_step   EXACTLY --steps pairs between two barriers, nothing but submit / wait in the timed region; before it
                        --warmup pairs per slot and at least 600 pairs in all run untimed (`warmup_pairs_run`), so that the
                        device clocks have followed the load;
  kernels               per-kernel device time, HIP events on the library's own stream, measured AFTER the timed region
                        on pairs run one at a time (what a rocprofv3 trace of `--streams 1` reproduces, profiles/);
  roofline(_fp64)       the dominant kernel of that pass against HBM (the metric asks for it) and against the FP64
                        vector peak (the bound that applies to this path);
  verified              the last pair of the timed region fetched and checked: the reference's known-answer hashes of
                        (x, y, index) of both edge lists and the pair / match counts of the workload; with the CPU
                        baseline also every edge, the sampled CSR rows and their NCC scores against the oracle's;
  with_h2d / with_h2d_d2h   the same loop with a NEW pair uploaded from host memory per step, and with the results
                        fetched back as well (PCIe-inclusive; never `value`);
  cpu_baseline          the CPU oracle (a port of the reference path) timed on this box's host cores, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from edge_based_visual_odometry_amd import sharding, synth  # noqa: E402  (numpy only: no GPU library is loaded here)


def Context(*a, **k):
    """edge_based_visual_odometry_amd.api.Context, imported on first use: the launcher branch of main() must start its
    ranks before this process has loaded the HIP library."""
    from edge_based_visual_odometry_amd.api import Context as C_
    return C_(*a, **k)

TOED_FLOPS_PER_PX = 37044            # 1,372 taps x 9 responses x 3 flops, as written in the reference
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E spec
FP64_VALU_PEAK_NOFMA_TF = 39.3       # 78.6 TFLOP/s vendor FP64 vector peak (FMA) / 2: mul and add are separate ops
FP64_VALU_SUSTAINED_FRAC = 4.0 / 4.7 # measured: a pure v_mul_f64 / v_add_f64 stream issues one wave instruction per 4.7
                                     # nominal cycles and SIMD, not per 4 (tools/ubench/bank_conflict.hip)

WORKLOADS = {
    # BASELINE.json configs[1] (the headline), configs[2] shape / calibration, configs[3] shape / calibration
    "kitti": dict(cfg="kitti", disparity=12,
                  label="configs[1]: single KITTI-shaped stereo pair 1241x376 (generator S2, scene 7+rank, 12 px "
                        "disparity), TOED both images (fp64, no FMA, bit-exact) + epipolar/disparity/orientation "
                        "candidate search + NCC, resident in HBM, replayed"),
    "euroc": dict(cfg="euroc", disparity=9, sequence=True,
                  label="configs[2]: EuRoC-shaped 752x480 sequence (SURVEY.md 8(d) config 3: scene 7, noise seeds (2k+1, 2k+2), "
                        "k px of global motion; 16 distinct frames resident in HBM, replayed for 64 steps), keyframe = frame 0; "
                        "per frame: cv::undistort of both images, TOED, candidate search + NCC, the whole stereo chain "
                        "(ebvo_stereo_finalize incl. the SIFT stages), temporal quads against the keyframe (grid + orientation "
                        "candidates, NCC on stored patches)"),
    "eth3d": dict(cfg="eth3d", disparity=9,
                  label="configs[3]: ETH3D delivery_area 942x489 stereo pair, same hot path (the rocprofv3 roofline run)"),
}
# known answers of the headline pair (S2 1241x376 scene 7, noise 1 / 2, shift 0 / 12)
MIN_WARM_PAIRS = 600   # untimed pairs before the timed region, whatever --warmup says (see main())
KITTI_KAT = dict(xyi_left="85fcd7aa12a47c8b", xyi_right="2b8a4c7b2e5454ca", n_left=126184, n_right=126340,
                 n_pairs=581657, n_matches=472947)


def algorithmic_bytes_per_pair(h, w, n_left, n_right, n_pairs):
    """SURVEY.md 8(d) 'Algorithmic bytes': 2P (u8 images in) + 32(NL+NR) edges out + 32(NL+NR) edges into matching
    + 8 Npairs (CSR) + 40 Npairs (4 sims + best)."""
    return 2 * h * w + 64 * (n_left + n_right) + 48 * n_pairs


# kernel id reported by the library's event profiler -> HIP kernel symbols bracketed by one event pair
KERNEL_SYMBOLS = {
    "toed_conv": {"strict": ["toed_conv_kernel"]},
    "toed_nms": {"strict": ["toed_nms_kernel"], "hybrid": ["toed_screen_fused_kernel"]},
    "toed_exact_centre": {"hybrid": ["toed_exact_centre_kernel"]},
    "toed_exact_mags": {"hybrid": ["toed_exact_mags_kernel", "toed_exact_decide_kernel"]},
    "cand_count": {"*": ["candidates_kernel<false>"]},
    "cand_fill": {"*": ["candidates_copy_kernel", "candidates_kernel<true>"]},
    "edge_patches": {"*": ["sincos_batch_kernel", "right_bank_kernel"]},
    "ncc_pairs": {"*": ["ncc_tile_kernel"]},
}


def sift_roofline():
    """VALU-issue roofline of the SIFT descriptor kernel (one wave per SIMD: its histograms fill the LDS) from the COMMITTED
    rocprofv3 counter passes of the KITTI-size chain (profiles/r04_chain_pmc_sq_chain.txt: launch duration, waves, VALU
    instructions per wave); not measured in this run.  Peak: 1,024 SIMDs x 2.4 GHz / 4 cycles per wave-instruction."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_chain_pmc_sq_chain.txt")
    try:
        for line in open(path):
            if line.startswith("sift_desc_pair_kernel ") or line.startswith("sift_desc_kernel "):
                f = line.split()
                avg_us, waves, valu = float(f[1]), float(f[2]), float(f[3])
                achieved = waves * valu / (avg_us * 1e-6) / 1e9
                peak = 1024 * 2.4e9 / 4 / 1e9
                return {"bound": "valu_issue", "kernel": f[0], "achieved": achieved, "peak": peak, "unit": "G wave-instructions/s",
                        "frac": achieved / peak, "waves": int(waves), "valu_instructions_per_wave": valu, "avg_launch_ms": avg_us * 1e-3,
                        "source": "profiles/r04_chain_pmc_sq_chain.txt (committed rocprofv3 SQ counter passes of the KITTI-size "
                                  "chain, both images in one launch; not measured in this run)"}
    except OSError:
        pass
    return None


def pmc_traffic(kernel_id, toed_mode):
    """HBM bytes per launch of a kernel id from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE,
    profiles/kernel_pmc_<mode>.json, written by tools/rocprof_summary.py) -- not measured in this run; None if that
    profile does not hold the kernel."""
    path = os.path.join(ROOT, "profiles", f"kernel_pmc_{toed_mode}.json")
    syms = KERNEL_SYMBOLS.get(kernel_id, {})
    syms = syms.get(toed_mode) or syms.get("*")
    if not syms or not os.path.exists(path):
        return None
    table = json.load(open(path)).get("kernels", {})
    total = 0.0
    for sym in syms:
        hit = [v for k, v in table.items() if k == sym or k.startswith(sym + "<")]
        if not hit:
            return None
        total += hit[0]["hbm_bytes_per_launch"]
    return total


def xyi_hash(edges) -> str:
    """FNV-1a-64 over the 16 raw bytes of (x, y) of every edge, then h ^= index; h *= prime (SURVEY.md 8(c) "xyi")."""
    h, prime, mask = 1469598103934665603, 1099511628211, (1 << 64) - 1
    xy = np.stack([edges["x"], edges["y"]], 1).astype("<f8").tobytes()
    idx = edges["index"].tolist()
    for k in range(len(idx)):
        for b in xy[16 * k:16 * k + 16]:
            h = ((h ^ b) * prime) & mask
        h = ((h ^ (idx[k] & mask)) * prime) & mask
    return f"{h:016x}"


def cpu_baseline(left, right, F):
    """The CPU oracle (a port of the reference's path) on this box's host cores, SURVEY.md 8(d) protocol: TOED = 1 warm-up + 5
    timed repetitions, median; the candidate search and the NCC pass UNSAMPLED, stage by stage as get_Stereo_Edge_Pairs runs
    them (src/Stereo_Matches.cpp:1374-1427): brute-force epipolar stage (OpenMP as written), disparity stage (its `#pragma omp
    for` is orphaned, :536: serial as written), orientation stage (parallel as written), NCC (orphaned pragma again, :573:
    serial as written) -- the two orphaned-pragma loops are timed both ways.  ~20-25 s on 16 cores.  Returns the timing record
    and everything it computed so that the GPU's results can be checked against it, pair by pair."""
    from tests import oracle as orc
    # a one-GPU box's CPU share is 16 cores; EBVO_CPU_THREADS overrides
    cores = int(os.environ.get("EBVO_CPU_THREADS", min(16, len(os.sched_getaffinity(0)))))

    def wall(fn, *a, **k):
        t0 = time.perf_counter()
        out = fn(*a, **k)
        return out, time.perf_counter() - t0

    # TOED of both images: warm-up + 5 repetitions
    reps = []
    for rep in range(6):
        (rl, rr), dt = wall(lambda: (orc.toed(left, math_mode=orc.LIBM, nthreads=cores), orc.toed(right, math_mode=orc.LIBM, nthreads=cores)))
        if rep:
            reps.append(dt)
    t_toed = float(np.median(reps))
    L, R = rl["edges"], rr["edges"]
    lines = orc.epipolar_lines(F, L)
    # stage 1: every left edge against every right edge (:91-109, :381-419) -- hundreds of candidates per edge
    (rp1, ci1), t_epi = wall(orc.epi_candidates, L, R, lines, stage_mask=orc.STAGE_EPIPOLAR, nthreads=cores)
    # stage 2 on those lists, serial as written and parallel as intended (:534-553)
    k_disp, t_disp_serial = wall(orc.filter_pairs, L, R, rp1, ci1, stage_mask=orc.STAGE_DISPARITY, nthreads=1)
    _, t_disp_par = wall(orc.filter_pairs, L, R, rp1, ci1, stage_mask=orc.STAGE_DISPARITY, nthreads=cores)
    rows1 = np.repeat(np.arange(len(L)), np.diff(rp1.astype(np.int64)))
    sel = k_disp.astype(bool)
    rp2 = np.concatenate([[0], np.cumsum(np.bincount(rows1[sel], minlength=len(L)))]).astype(np.int32)
    ci2 = ci1[sel]
    # stage 3 on the survivors (:863-915, a parallel region as written)
    k_or, t_orient = wall(orc.filter_pairs, L, R, rp2, ci2, stage_mask=orc.STAGE_ORIENTATION, nthreads=cores)
    rows2 = rows1[sel]
    sel3 = k_or.astype(bool)
    rp3 = np.concatenate([[0], np.cumsum(np.bincount(rows2[sel3], minlength=len(L)))]).astype(np.int32)
    ci3 = ci2[sel3]
    # NCC of every surviving pair with the left patches (:555-616), serial as written and parallel as intended
    cand = R[ci3]
    _, t_ncc_serial = wall(orc.ncc_pairs, left, right, L, cand, rp3, math_mode=orc.LIBM, nthreads=1)
    libm_ncc, t_ncc_par = wall(orc.ncc_pairs, left, right, L, cand, rp3, math_mode=orc.LIBM, nthreads=cores)
    # what the baseline computed, in the reference's own arithmetic (glibc atan2 / sin / cos): kept for the check after it
    libm = dict(left=L, right=R, stage1=(rp1, ci1), stage2=(rp2, ci2), stage3=(rp3, ci3), sims=libm_ncc[0], best=libm_ncc[1],
                keep=libm_ncc[2])
    # one thread throughout: TOED of one image timed once (x 2); the brute-force stage scaled from every 64th left edge
    _, t1 = wall(orc.toed, left, math_mode=orc.LIBM, nthreads=1)
    s1 = 64
    _, te1 = wall(orc.epi_candidates, L[::s1], R, lines[::s1], stage_mask=orc.STAGE_EPIPOLAR, nthreads=1)
    t_one = 2.0 * t1 + te1 * s1 + t_disp_serial + t_orient * cores + t_ncc_serial      # (orientation stage: scaled from the parallel run)
    as_written = t_toed + t_epi + t_disp_serial + t_orient + t_ncc_serial
    parallel = t_toed + t_epi + t_disp_par + t_orient + t_ncc_par
    # checker leg (not timed): the same lists in the arithmetic the GPU path uses (shared sin / cos / atan2)
    Lp, Rp = orc.toed(left, nthreads=cores)["edges"], orc.toed(right, nthreads=cores)["edges"]
    same_xy = len(Lp) == len(L) and len(Rp) == len(R) and all(
        np.array_equal(a[f], b[f]) for a, b in ((Lp, L), (Rp, R)) for f in ("x", "y", "index"))
    sims, _, keep, _ = orc.ncc_pairs(left, right, Lp, Rp[ci3], rp3, nthreads=cores)
    record = {
        "value": 1.0 / parallel, "unit": "stereo pairs/s", "cores": cores, "kind": "port",
        "value_as_written": 1.0 / as_written,
        "value_1_thread": 1.0 / t_one,
        "seconds_per_pair": parallel, "seconds_per_pair_as_written": as_written, "seconds_per_pair_1_thread": t_one,
        "seconds": {"toed_both_images_median_of_5": t_toed, "toed_repetitions": reps, "epipolar_stage_brute_force": t_epi,
                    "disparity_stage_serial_as_written": t_disp_serial, "disparity_stage_parallel": t_disp_par,
                    "orientation_stage": t_orient, "ncc_serial_as_written": t_ncc_serial, "ncc_parallel": t_ncc_par},
        "pairs": {"after_epipolar_stage": int(len(ci1)), "after_disparity_stage": int(len(ci2)), "after_orientation_stage": int(len(ci3))},
        "sample": (f"oracle/ (C + OpenMP restatement of the reference path, gcc -O2, no FMA, {cores} threads) on the same S2 pair, "
                   "UNSAMPLED: TOED of both images (1 warm-up + 5 repetitions, median), then the stages of get_Stereo_Edge_Pairs one "
                   "after the other on every left edge: brute-force epipolar stage, disparity stage, orientation stage, NCC with the "
                   "left patches.  `value` counts the disparity and NCC loops in parallel (what their pragmas intend); "
                   "`value_as_written` counts them serially (what the reference executes: both pragmas are orphaned, "
                   "src/Stereo_Matches.cpp:536, :573)"),
    }
    return record, dict(left=Lp, right=Rp, stride=1, row_ptr=rp3, col_idx=ci3, sims=sims, keep=keep, xy_equal_libm=bool(same_xy),
                        libm=libm)


def reference_arithmetic_report(ctx, out, libm, F):
    """The fetched results of the timed pair against the CPU baseline's own (LIBM) results: see tests/reference_arithmetic.py.
    Stages 1 and 2 (epipolar; epipolar + disparity) are produced here by the library's host-buffer search with the
    reference's stage masks; stage 3, the scores and keep are what the timed pipeline returned."""
    from tests import reference_arithmetic as ra
    lines = ctx.epipolar_lines(F, out["left"])
    gpu = dict(left=out["left"], right=out["right"], stage3=(out["row_ptr"], out["col_idx"]), sims=out["sims"], best=out["best"],
               keep=out["keep"])
    gpu["stage1"] = ctx.epi_candidates(out["left"], out["right"], lines, stage_mask=1)
    gpu["stage2"] = ctx.epi_candidates(out["left"], out["right"], lines, stage_mask=3)
    return ra.compare(libm, gpu)


def check_against_oracle(out, chk):
    """Every edge of both images, the WHOLE candidate CSR, all four NCC scores and the keep flag of every pair, bit for bit."""
    def same(a, b):
        return a.shape == b.shape and bool(((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all())
    if not chk.get("xy_equal_libm", True):
        return "the oracle's edge positions differ between its libm and shared-math modes"
    for side in ("left", "right"):
        for f in ("x", "y", "theta"):
            if not same(out[side][f].copy(), chk[side][f].copy()):
                return f"{side}.{f} differs from the oracle"
        if not (out[side]["index"] == chk[side]["index"]).all():
            return f"{side}.index differs from the oracle"
    if not np.array_equal(out["row_ptr"], chk["row_ptr"]):
        return "candidate row offsets differ from the oracle"
    if not np.array_equal(out["col_idx"], chk["col_idx"]):
        return "candidate indices differ from the oracle"
    if not same(np.ascontiguousarray(out["sims"]), np.ascontiguousarray(chk["sims"])):
        return "NCC scores differ from the oracle"
    if not np.array_equal(out["keep"], chk["keep"]):
        return "NCC keep flags differ from the oracle"
    return None


# what the timing ids of `kernels` bracket (the ids are stages, HIP events around the launches of a stage)
KERNEL_GROUPS = {
    "hybrid": {"toed_nms": "toed_screen_fused",
               "toed_compact": "toed_compact_phase + toed_need_count / _compact (row offsets summed by the compaction blocks)",
               "toed_exact_centre": "toed_exact_centre", "toed_exact_mags": "toed_exact_mags + toed_exact_decide",
               "toed_finalize": "toed_cand_scatter (ranks from per-chunk counts)", "scan": "scan_reduce + scan_apply (row_ptr)",
               "cand_boxes": "match_prep (lines + boxes + sincos + row_pairs in one launch)", "cand_count": "candidates<count>",
               "cand_fill": "candidates<fill> (with the copy of the staged rows as its prologue)", "edge_patches": "right_bank",
               "ncc_pairs": "ncc_tile + pair_result"},
    "strict": {"toed_conv": "toed_conv", "toed_nms": "toed_nms", "toed_rowscan": "toed_rowscan", "toed_compact": "toed_compact",
               "toed_finalize": "toed_finalize", "scan": "scan_reduce + scan_apply",
               "cand_boxes": "match_prep (lines + boxes + sincos + row_pairs in one launch)",
               "cand_count": "candidates<count>", "cand_fill": "candidates<fill> (with the copy of the staged rows as its prologue)",
               "edge_patches": "right_bank", "ncc_pairs": "ncc_tile + pair_result"},
}


def dropin_loop(ctx, params, pool, calib, nslots, steps, chains=2):
    """What a frame loop over StereoMatcherHIP::stereo_edge_pairs_begin / _end does per frame (include/ebvo/adapters.hpp; the
    one-pass body of get_Stereo_Edge_Pairs, integration/stereo_matches_hip.cpp): a NEW pair uploaded from host memory, TOED +
    candidates + NCC, then the later stages on the device with the SIFT filter (ebvo_stereo_finalize_submit / _wait) and the
    final pairs + output rows copied back.  `nslots` frames in flight, up to `chains` of them in their later stages: the
    chain of a frame is enqueued without a host synchronisation, so the chains of different slots overlap on the device."""
    t0 = time.perf_counter()
    sub = done = 0
    n_final = 0
    in_chain = []                                   # slots whose chain is enqueued, oldest first

    def launch(k):
        nonlocal sub
        if sub < steps:
            ctx.stereo_upload(*pool[sub % len(pool)], slot=k)
            ctx.stereo_submit(params, slot=k)
            sub += 1

    def retire():
        nonlocal done, n_final
        k = in_chain.pop(0)
        fc, fin = ctx.stereo_finalize_wait(slot=k)
        n_final += fc["n_final"]
        done += 1
        launch(k)

    for k in range(min(nslots, steps)):
        launch(k)
    entered = 0
    while entered < steps:
        k = entered % nslots
        ctx.stereo_wait(slot=k)
        ctx.stereo_finalize_submit(calib, slot=k, use_sift=True)
        in_chain.append(k)
        entered += 1
        while len(in_chain) >= max(1, min(chains, nslots)):
            retire()
    while in_chain:
        retire()
    return time.perf_counter() - t0, n_final / max(1, steps)


def frame_loop(ctx, params, pool, nslots, steps, upload, fetch):
    """`steps` pairs with `nslots` in flight; upload: a NEW pair from host memory per step; fetch: None, or the selection
    (ebvo_hip.h EBVO_FETCH_*) copied back through the slot's page-locked staging (ebvo_stereo_fetch_begin / _end: the copy
    of pair k runs while the other slots compute; slot k is resubmitted once its results have been read), or "pageable":
    every array copied synchronously into fresh numpy arrays (ebvo_stereo_fetch).  Returns (seconds, bytes per pair)."""
    t0 = time.perf_counter()
    sub = done = 0
    nbytes = 0

    def launch(k):
        nonlocal sub
        if sub < steps:
            if upload:
                ctx.stereo_upload(*pool[sub % len(pool)], slot=k)
            ctx.stereo_submit(params, slot=k)
            sub += 1

    def consume(k):
        nonlocal nbytes
        out = ctx.stereo_fetch_end(slot=k)
        nbytes += sum(v.nbytes for v in out.values() if v is not None)
        if out["keep"] is not None and int(out["keep"][-1]) > 1:      # touch the data
            raise RuntimeError("keep flag out of range")

    for k in range(min(nslots, steps)):
        launch(k)
    pending = None
    while done < steps:
        k = done % nslots
        cnt = ctx.stereo_wait(slot=k)
        done += 1
        if fetch == "pageable":
            out = ctx.stereo_fetch(cnt, slot=k)
            nbytes += sum(v.nbytes for v in out.values() if v is not None)
            launch(k)
        elif fetch and nslots < 2:
            # one slot: nothing to overlap the copy with, and the slot cannot be waited on again before it is resubmitted
            ctx.stereo_fetch_begin(slot=k, what=fetch)
            consume(k)
            launch(k)
        elif fetch:
            ctx.stereo_fetch_begin(slot=k, what=fetch)
            if pending is not None:
                consume(pending)
                launch(pending)
            pending = k
        else:
            launch(k)
    if pending is not None:
        consume(pending)
    return time.perf_counter() - t0, nbytes / max(1, steps)


def ingest_loop(ctx, params, ring, nslots, steps, fetch=None, host_us=None):
    """A streamed sequence at the resident rate (round 4): `nslots` pairs in flight on `nslots + 1` slots; the images of step
    i + 1 are uploaded (ebvo_stereo_upload_async: DMA from the page-locked frame ring on the context's upload stream, the host
    does not wait) while step i is still being matched, so that a submission never waits for its images -- what a frame loop
    that reads frame k + 1 while frame k is matched does (src/Pipeline.cpp:77-99, cmd/main_VO.cpp:99-113).  fetch: None, or
    "compact" (ebvo_stereo_fetch_compact_begin / _end: (x, y) pairs, CSR, best, keep bits through page-locked staging, the
    copy overlaps the other slots' kernels), "push" (the same arrays written by the pair's own chain into page-locked host
    memory: `params` must carry EBVO_PAIR_PUSH), or an EBVO_FETCH_* selection.  Requires ctx.set_slots(nslots + 1) or more.
    Returns (seconds, bytes fetched per pair)."""
    S = nslots + 1
    t0 = time.perf_counter()
    nbytes = 0
    uploaded = submitted = done = 0

    def upload(slot):
        nonlocal uploaded
        t_ = time.perf_counter()
        ctx.stereo_upload_async(*ring[uploaded % len(ring)], slot=slot)
        if host_us is not None:
            host_us["upload_async"] = host_us.get("upload_async", 0.0) + (time.perf_counter() - t_) * 1e6 / steps
        uploaded += 1

    def timed(name, fn, *a, **k):
        t_ = time.perf_counter()
        r = fn(*a, **k)
        if host_us is not None:
            host_us[name] = host_us.get(name, 0.0) + (time.perf_counter() - t_) * 1e6 / steps
        return r

    def begin(slot):
        if fetch == "push":                       # nothing to start: the pair's own chain has written the results to host memory
            return
        if fetch == "compact":
            ctx.stereo_fetch_compact_begin(slot=slot)
        else:
            ctx.stereo_fetch_begin(slot=slot, what=fetch)

    def consume(slot):
        nonlocal nbytes
        out = (ctx.stereo_pushed_view(slot=slot) if fetch == "push" else
               ctx.stereo_fetch_compact_end(slot=slot) if fetch == "compact" else ctx.stereo_fetch_end(slot=slot))
        nbytes += sum(v.nbytes for v in out.values() if isinstance(v, np.ndarray))
        if fetch in ("compact", "push"):
            if int(out["keep_bits"][0]) < 0 or out["n_matches"] > out["n_pairs"]:      # touch the data
                raise RuntimeError("compact results out of range")
        elif out["keep"] is not None and int(out["keep"][-1]) > 1:
            raise RuntimeError("keep flag out of range")

    for k in range(min(nslots, steps)):
        upload(k)
        ctx.stereo_submit(params, slot=k)
        submitted += 1
    ahead = None                                   # the slot whose images are uploaded and not yet submitted
    if uploaded < steps:
        ahead = uploaded % S
        upload(ahead)
    pending = None
    while done < steps:
        k = done % S
        timed("wait", ctx.stereo_wait, slot=k)
        done += 1
        if fetch == "push":
            consume(k)                             # already in host memory
        elif fetch:
            begin(k)                               # k's results start travelling; read at the next turn
            if pending is not None:
                consume(pending)                   # ... before `pending` (= ahead) is submitted again
            pending = k
        if ahead is not None:                      # its upload was enqueued a whole pair earlier
            timed("submit", ctx.stereo_submit, params, slot=ahead)
            submitted += 1
            ahead = None
        if uploaded < steps:                       # the images the slot will be submitted with at the next turn
            ahead = k
            upload(k)
    if pending is not None:
        consume(pending)
    return time.perf_counter() - t0, nbytes / max(1, steps)


def boundary_bench_cpp(left, right, H, W, iters=10, toed_mode="hybrid"):
    """Builds and runs tools/boundary_bench.cpp (g++ against include/ebvo/adapters.hpp and the in-tree library) as a child
    process; returns its JSON object, or None when there is no compiler / the build or the run fails."""
    import subprocess
    import tempfile
    from edge_based_visual_odometry_amd import _lib
    try:
        libdir = os.path.dirname(_lib.LIB_PATH)
        with tempfile.TemporaryDirectory() as tmp:
            exe = os.path.join(tmp, "boundary_bench")
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                                   os.path.join(ROOT, "tools", "boundary_bench.cpp"), "-o", exe, "-L", libdir, "-lebvo_hip",
                                   f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL, timeout=120)
            lp, rp_ = os.path.join(tmp, "l.raw"), os.path.join(tmp, "r.raw")
            np.ascontiguousarray(left).tofile(lp)
            np.ascontiguousarray(right).tofile(rp_)
            out = subprocess.run([exe, lp, rp_, str(H), str(W), str(iters)], capture_output=True, timeout=120, check=True,
                                 env=dict(os.environ, EBVO_TOED_MODE=toed_mode))
            return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001 - the Python leg stands in
        return None


def host_threads(args):
    """--host-threads "2,4": the thread counts of the optional multi-context legs (none by default)"""
    return [int(x) for x in args.host_threads.split(",") if x.strip()]


def threaded_dropin(T, args, H, W, F, device, pool, calib, steps):
    """frames/s of dropin_loop run by T host threads, each on its own context (three slots each)."""
    import threading
    ctxs = []
    for t in range(T):
        c = Context(H, W, device=device, toed_mode=args.toed_mode)
        c.set_slots(3)
        p = c.default_params(F)
        dropin_loop(c, p, pool, calib, 3, 3)     # untimed: sizes the buffers, one context after the other
        ctxs.append((c, p))
    start = threading.Barrier(T + 1)
    per = [steps // T + (1 if t < steps % T else 0) for t in range(T)]

    def work(t):
        c, p = ctxs[t]
        start.wait()
        dropin_loop(c, p, pool, calib, 3, per[t])

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(1e-4)                  # a thread back from a C call should not wait 5 ms for the interpreter lock
    for x in th:
        x.start()
    start.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    sys.setswitchinterval(old_switch)
    for c, _ in ctxs:
        c.close()
    return steps / dt


def threaded_sequence(T, args, H, W, F, device, frames, calib, cal, steps):
    """frames/s of the EuRoC frame loop with T contexts, each driven by its own host thread: context t holds the keyframe
    (frame 0) and the frames k = t (mod T); ctypes calls release the interpreter lock, so the chains run concurrently."""
    import threading
    ctxs, slots_of = [], []
    for t in range(T):
        mine = [k for k in range(len(frames)) if k % T == t]
        c = Context(H, W, device=device, toed_mode=args.toed_mode)
        c.set_slots(len(mine) + 1)
        if "dist" in cal:
            c.set_undistort(cal["K"], cal["dist"], cal["K_right"], cal["dist_right"])
        p = c.default_params(F)
        c.stereo_upload(*frames[0], slot=0)
        c.stereo_submit(p, slot=0)
        c.stereo_wait(slot=0)
        c.stereo_finalize(calib, slot=0, use_sift=True)
        c.temporal_set_keyframe(slot=0)
        for j, k in enumerate(mine):
            c.stereo_upload(*frames[k], slot=j + 1)
        ctxs.append((c, p))
        slots_of.append(list(range(1, len(mine) + 1)))

    def one(c, p, slot):
        c.stereo_submit(p, slot=slot)
        c.stereo_wait(slot=slot)
        c.stereo_finalize(calib, slot=slot, use_sift=True)
        c.temporal_match(slot=slot, fetch=False, stages=0)

    for (c, p), sl in zip(ctxs, slots_of):       # untimed: sizes every buffer, one context after the other
        for slot in sl[:2]:
            one(c, p, slot)
    start = threading.Barrier(T + 1)
    per = [steps // T + (1 if t < steps % T else 0) for t in range(T)]

    def work(t):
        c, p = ctxs[t]
        start.wait()
        for i in range(per[t]):
            one(c, p, slots_of[t][i % len(slots_of[t])])

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(1e-4)                  # a thread back from a C call should not wait 5 ms for the interpreter lock
    for x in th:
        x.start()
    start.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    sys.setswitchinterval(old_switch)
    for c, _ in ctxs:
        c.close()
    return steps / dt


def verify_sequence_frame(ctx, frames, k, params, calib, cal, H, W, F=None):
    """After the timed region of the sequence workload: frame k once more, its results fetched and compared with the CPU oracle
    -- both TOED edge lists (on the undistorted images), every candidate quad against the keyframe, both NCC maxima and the
    keep flag of every quad, and every final quad of the whole temporal chain (tests/oracle_chain.py: temporal_reference).
    Returns the list of differences (empty = verified)."""
    from tests import oracle as orc
    from tests import oracle_chain

    def triple(imgs):
        if "dist" not in cal:
            return imgs[0], imgs[0], imgs[1]
        return imgs[0], orc.undistort(imgs[0], cal["K"], cal["dist"]), orc.undistort(imgs[1], cal["K_right"], cal["dist_right"])

    def mates(slot):
        ctx.stereo_submit(params, slot=slot)
        c = ctx.stereo_wait(slot=slot)
        out = ctx.stereo_fetch(c, slot=slot)
        _, fin = ctx.stereo_finalize(calib, slot=slot, use_sift=True)
        return out, out["left"][fin["left_index"]], fin["right"]

    problems = []
    _, kfL, kfR = mates(0)                          # the keyframe's mates (slot 0 still holds frame 0)
    out, cfL, cfR = mates(k)
    counts, q = ctx.temporal_match(slot=k, fetch=True, stages=1)
    t0, tk = triple(frames[0]), triple(frames[k])
    for side, img in (("left", tk[1]), ("right", tk[2])):
        ref = orc.toed(img)["edges"]
        got = out[side]
        if len(ref) != len(got) or any((ref[f].view(np.uint64) != got[f].view(np.uint64)).any() for f in ("x", "y", "theta")):
            problems.append(f"frame {k}: {side} edge list differs from the oracle")
    ref = oracle_chain.temporal_reference(kfL, kfR, cfL, cfR, t0, tk, W, H)
    problems += [f"frame {k}: {p}" for p in oracle_chain.temporal_problems(counts, q, ref)]
    ra_rep = None
    if F is not None:
        # the frame's stereo stage against the oracle in the reference's OWN arithmetic (glibc atan2 / sin / cos): detector on
        # the undistorted images, NCC on the raw ones (tests/reference_arithmetic.py)
        from tests import reference_arithmetic as ra
        libm = ra.oracle_libm_stages(tk[1], tk[2], frames[k][0], frames[k][1], F, cores=ra.default_cores())
        ra_rep = reference_arithmetic_report(ctx, out, libm, F)
        problems += [f"frame {k}: {key} is false: {ra_rep['flips'][:4]}" for key in ra.BOOLEANS if not ra_rep[key]]
    return problems, dict(reference_arithmetic=ra_rep, frame=k, n_kf=counts["n_kf"], n_cf=counts["n_cf"], candidate_quads=counts["n_candidates"],
                          quads_kept=counts["n_kept"], final_quads=counts["n_final"])


def sequence_bench(args, wl, H, W, F, device, rank, world, dist, reduce_device):
    """configs[2]: a sequence with a keyframe.  A step = one frame through the whole per-frame path; the frames are resident
    (one slot each), so the timed region holds no host-to-device image traffic, like the headline workload."""
    import torch
    cal = synth.CALIB[wl["cfg"]]
    n_frames = 16
    frames = []
    for k in range(n_frames):
        l, r = synth.stereo_pair("s2", H, W, scene=7 + rank, noise_base=100 * rank + 2 * k, disparity=wl["disparity"])
        frames.append((np.roll(l, k, axis=1), np.roll(r, k, axis=1)))
    kl = [cal["K"][0], 0, cal["K"][2], 0, cal["K"][1], cal["K"][3], 0, 0, 1]
    kr = [cal["K_right"][0], 0, cal["K_right"][2], 0, cal["K_right"][1], cal["K_right"][3], 0, 0, 1]
    calib = (kl, kr, cal["R21"], cal["T21"])
    ctx = Context(H, W, device=device, toed_mode=args.toed_mode)
    ctx.set_slots(n_frames)
    if "dist" in cal:
        ctx.set_undistort(cal["K"], cal["dist"], cal["K_right"], cal["dist_right"])
    params = ctx.default_params(F)
    for k, (l, r) in enumerate(frames):
        ctx.stereo_upload(l, r, slot=k)

    def frame(k, first=False, stages=0):
        """one frame, every stage waited for before the next is enqueued.  stages = 0: temporal quads through the NCC filter, what
        BASELINE configs[2] names; 1: every stage of get_Temporal_Edge_Pairs_from_Quads (SIFT filter, both Best-Nearly-Best
        tests, refinement of both cameras, clustering)"""
        ctx.stereo_submit(params, slot=k)
        c = ctx.stereo_wait(slot=k)
        fc, _ = ctx.stereo_finalize(calib, slot=k, use_sift=True)
        if first:
            ctx.temporal_set_keyframe(slot=k)
            return c, fc, None
        tc, _ = ctx.temporal_match(slot=k, fetch=False, stages=stages)
        return c, fc, tc

    DEFAULT_LAGS = (1, 3, 4)   # steps between a frame's A and its B, C, D: five frames in flight (tools/gpu_pipeline_hosttime.py:
                               # 957 / 1009 / 1020 frames/s for (1, 2, 3) / (1, 3, 4) / (1, 4, 5); round 3 ran (1, 2, 3))

    def frame_pipeline(slots, stages=0, lags=None):
        """The frames `slots` (one per step, each resident in its slot) as a software pipeline over four enqueue-only steps --
        A: TOED + candidates + NCC submitted; B: its counts read, the stereo chain enqueued (ebvo_stereo_finalize_submit); C: the
        chain's counts read, the temporal stages enqueued (ebvo_temporal_match_submit); D: their counts read -- so the host
        never waits for the frame it has just enqueued and the chains of up to five frames overlap on the device.  Returns the
        per-frame (stereo counts, chain counts, temporal counts), in order."""
        n = len(slots)
        LAG_B, LAG_C, LAG_D = lags or DEFAULT_LAGS
        res = [[None, None, None] for _ in range(n)]
        for i in range(n + LAG_D):
            if i < n:
                assert slots[i] not in slots[max(0, i - LAG_D):i], "a slot re-enters the pipeline before it has left it"
                ctx.stereo_submit(params, slot=slots[i])                                      # A
            if 0 <= i - LAG_B < n:
                res[i - LAG_B][0] = ctx.stereo_wait(slot=slots[i - LAG_B])                    # B
                ctx.stereo_finalize_submit(calib, slot=slots[i - LAG_B], use_sift=True)
            if 0 <= i - LAG_C < n:
                res[i - LAG_C][1] = ctx.stereo_finalize_wait(slot=slots[i - LAG_C], fetch=False)[0]   # C
                ctx.temporal_match_submit(slot=slots[i - LAG_C], stages=stages)
            if 0 <= i - LAG_D < n:
                res[i - LAG_D][2] = ctx.temporal_match_wait(slot=slots[i - LAG_D], fetch=False)[0]    # D
        return [tuple(r) for r in res]

    frame(0, first=True)                               # keyframe = frame 0 (src/Pipeline.cpp:133-138)
    for k in range(min(args.warmup, n_frames)):
        frame(k)
    for _ in range(3):                                 # untimed: sizes the quad buffers of every slot, brings the clocks up, and
        frame_pipeline(list(range(n_frames)))          # gives every slot the three submissions after which its pair chain is a graph

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    totals = dict(pairs=0, matches=0, final=0, quads=0, kept=0)
    for c, fc, tc in frame_pipeline([step % n_frames for step in range(args.steps)]):
        totals["pairs"] += c.n_pairs
        totals["matches"] += c.n_matches
        totals["final"] += fc["n_final"]
        totals["quads"] += tc["n_candidates"]
        totals["kept"] += tc["n_kept"]
    barrier()
    dt = time.perf_counter() - t0
    per_rank = sharding.gather_over_ranks(args.steps / dt, dist, reduce_device)
    dt = sharding.max_over_ranks(dt, dist, reduce_device)
    prof = None
    if rank == 0:
        ctx.profile_reset()
        ctx.profile_enable(True, every=1)
        for k in range(min(4, n_frames)):
            frame(k)
        ctx.profile_enable(False)
        prof = ctx.profile_get()
        n_ser = min(4, n_frames)
        kernels = {k: {"ms_per_step": v[0] / n_ser, "launches_per_step": v[1] / n_ser} for k, v in prof.items() if v[1]}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        c, fc, tc = frame(1)
        # the same loop with every stage of the temporal chain (never `value`: the config names the NCC quads)
        # (64 frames behind an untimed pass over the 16 slots: a 16-frame burst spent a fifth of its time filling and draining
        # the pipeline -- 525-550 frames/s where tools/gpu_pipeline_hosttime.py euroc 1 measures 670-730 over 96 frames)
        n_full = 64
        frame(1, stages=1)
        frame_pipeline(list(range(n_frames)), stages=1)
        t1 = time.perf_counter()
        tcf = frame_pipeline([k % n_frames for k in range(n_full)], stages=1)[-1][2]
        t_full = (time.perf_counter() - t1) / n_full
        # ... and frame after frame, every stage waited for (what the loop was before the enqueue-only chains)
        n_serial = min(16, n_frames)
        t1 = time.perf_counter()
        for k in range(n_serial):
            frame(k % n_frames)
        t_serial = (time.perf_counter() - t1) / n_serial
        # the same frames through T contexts driven by T host threads (the library's model: one ebvo_ctx per host thread):
        # a frame is a host-sequenced chain of ~180 short launches, so several chains share the device well
        threaded = {}
        for T in host_threads(args):
            threaded[str(T)] = threaded_sequence(T, args, H, W, F, device, frames, calib, cal, min(args.steps, 64))
        alg_bytes = algorithmic_bytes_per_pair(H, W, c.n_left, c.n_right, c.n_pairs)
        dom_s = kernels[dom]["ms_per_step"] * 1e-3 / max(1.0, kernels[dom]["launches_per_step"])
        result = {
            "metric": f"stereo frames/sec (undistort + TOED + stereo chain + temporal NCC) on {wl['cfg']} {W}x{H}; achieved HBM GB/s",
            "value": sharding.job_throughput(world, args.steps, dt), "unit": "stereo frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "per_rank_frames_per_s": per_rank,
            **sharding.rank_balance(per_rank, world, sharding.job_throughput(world, args.steps, dt)),
            "config": {"workload": wl["label"], "shape": f"{W}x{H}", "toed_mode": args.toed_mode,
                       "edges_left": c.n_left, "edges_right": c.n_right, "candidate_pairs": c.n_pairs, "ncc_matches": c.n_matches,
                       "final_stereo_mates": fc["n_final"], "temporal_candidate_quads": tc["n_candidates"],
                       "temporal_quads_kept": tc["n_kept"], "per_step_averages": {k: v / args.steps for k, v in totals.items()},
                       "parallelism": f"{world} independent sequence(s), one per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": alg_bytes / dom_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / dom_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_s * 1e3,
                         "note": "the stereo hot path's algorithmic bytes (SURVEY.md 8(d)) over the dominant kernel id's launch "
                                 "duration (HIP events, frames one at a time, after the timed region)"},
            "kernels": kernels,
            "roofline_sift": sift_roofline(),
            "frames_per_s_by_host_threads": threaded,
            "full_temporal_chain_frames_per_s": 1.0 / t_full,
            "one_frame_at_a_time_frames_per_s": 1.0 / t_serial,
            "pipeline_note": "value: the frame loop as a software pipeline of enqueue-only steps (TOED + matching | stereo chain | "
                             "temporal stages | counts), up to five frames in flight in their own slots from one host thread and "
                             "one context; one_frame_at_a_time: every stage waited for before the next is enqueued",
            "full_temporal_chain_note": "64 frames behind an untimed pass over the 16 slots: the same frame loop with the rest of get_Temporal_Edge_Pairs_from_Quads after the NCC "
                                        "filter (SIFT filter, Best-Nearly-Best on NCC and SIFT scores, photometric refinement "
                                        "of both cameras, edge clustering): %d final quads per frame" % tcf["n_final"],
        }
        if not args.no_verify:
            problems, what = verify_sequence_frame(ctx, frames, min(3, n_frames - 1), params, calib, cal, H, W, F)
            result["verified"] = not problems
            ra_rep = what.pop("reference_arithmetic")
            if ra_rep is not None:
                from tests.reference_arithmetic import BOOLEANS
                result["reference_arithmetic"] = ra_rep
                result.update({key: ra_rep[key] for key in BOOLEANS})
            result["verified_what"] = dict(what, note="both edge lists, every candidate quad (row_ptr / col_idx), both NCC maxima "
                                                      "and the keep flag of every quad, every final quad of the temporal chain: bit "
                                                      "for bit against the CPU oracle (tests/oracle_chain.py)")
            if problems:
                result["verification_failures"] = problems
        print(json.dumps(result))
        if result.get("verified") is False:
            ctx.close()
            sys.exit(3)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node, one sequence each.  N > 1 with no WORLD_SIZE in the environment: bench.py "
                         "starts the N ranks itself; under torch.distributed.run it must equal WORLD_SIZE")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="kitti", choices=sorted(WORKLOADS),
                    help="kitti = the headline (BASELINE.json configs[1]); euroc / eth3d = the other reference shapes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the result check after the timed region (profiling runs)")
    ap.add_argument("--no-transfer-legs", action="store_true", help="skip the with_h2d / with_h2d_d2h loops")
    ap.add_argument("--store-sims", action="store_true",
                    help="the timed pipeline also stores the four similarities of every pair (32 of the 41 bytes the NCC stage writes "
                         "per pair; round 3's behaviour).  Default: final score + keep flag only, as the reference keeps them")
    ap.add_argument("--no-ingest", action="store_true", help="skip value_with_h2d (the timed loop fed from a page-locked frame ring)")
    ap.add_argument("--toed-mode", default="hybrid", choices=["strict", "hybrid"],
                    help="strict: direct-form convolution at every pixel; hybrid: separable screen + exact "
                         "re-evaluation of the candidates (bit-identical edges, ~3x less work)")
    ap.add_argument("--dist-backend", default="auto",
                    help="torch.distributed backend for the barrier / max: nccl (= RCCL), gloo, or auto = nccl when every rank "
                         "has its own GPU, gloo when N ranks rehearse on fewer GPUs")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="run only the multi-rank plumbing (launcher, rendezvous, barrier, reductions) and stop before any GPU call")
    ap.add_argument("--streams", type=int, default=6,
                    help="stereo pairs kept in flight per GPU (slots; one HIP stream each up to 3, from 4 on their kernels are "
                         "dealt to min(4, slots - 1) streams of the context)")
    ap.add_argument("--host-threads", default="",
                    help="e.g. 2,4,8: also run the frame loops (euroc: the sequence; kitti / eth3d: the one-pass drop-in) in that "
                         "many host threads, one context each -- extra keys of the JSON line, never `value`")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: this process becomes the launcher (before it has touched the GPU or loaded the HIP
    # library) -- N fresh child processes, one per GPU, each re-entering main() with RANK / WORLD_SIZE set ---------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(sharding.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    info = sharding.rank_info()
    rank, local_rank, world = info.rank, info.local_rank, info.world
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks", file=sys.stderr)
        sys.exit(2)
    dist = None
    # every rank on its own cores (its GPU's NUMA node when sysfs says which), thread pools capped: before torch starts its own
    n_dev_sysfs = len([d for d in os.listdir("/sys/class/drm") if d.startswith("renderD")]) if os.path.isdir("/sys/class/drm") else 0
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    rank_cpus = sharding.pin_rank(local_rank, local_world, max(1, n_dev_sysfs)) if world > 1 and not os.environ.get("EBVO_NO_PIN") else None
    import torch
    ndev = torch.cuda.device_count()             # counting devices does not initialise the GPU
    device = local_rank % max(1, ndev)           # one rank per GPU on a full node; wraps only when rehearsing N > #GPUs
    backend = args.dist_backend
    if backend == "auto":                        # RCCL needs one device per rank; a rehearsal of N ranks on fewer GPUs uses gloo
        backend = "nccl" if 0 < world <= ndev else "gloo"
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    reduce_device = f"cuda:{device}" if backend == "nccl" else "cpu"

    if args.selftest_launch:
        # the N > 1 plumbing only (launcher, rendezvous, barrier, MAX / gather over ranks), no GPU call: what
        # tests/test_bench_launch.py drives with two gloo ranks on a machine without a GPU
        if dist is not None:
            dist.barrier()
        mine = 0.25 * (rank + 1)
        per_rank = sharding.gather_over_ranks(mine, dist, reduce_device)
        dt = sharding.max_over_ranks(mine, dist, reduce_device)
        cpus = sharding.gather_int_lists(sorted(os.sched_getaffinity(0)), 1024, dist, reduce_device)
        if rank == 0:
            print(json.dumps({"selftest_launch": True, "n_gpus": world, "backend": backend if world > 1 else None,
                              "per_rank_seconds": per_rank, "max_seconds": dt, "rank_cpus": cpus, "pinned": rank_cpus is not None,
                              "omp_num_threads": os.environ.get("OMP_NUM_THREADS"),
                              "value": sharding.job_throughput(world, args.steps, dt),
                              "sequences": [sharding.rank_workload(r) for r in range(world)]}))
        if dist is not None:
            dist.destroy_process_group()
        return

    # The timed loops are driven from Python: a generation-2 collection in a process that has imported torch walks its whole
    # object graph (40-70 ms, seen as one call of a 300-pair loop "taking" 130 us per pair).  Collect once, freeze what exists,
    # and keep the collector off while timing -- the harness's pauses are not the library's.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    wl = WORKLOADS[args.workload]
    H, W = synth.SHAPES[wl["cfg"]]
    # one sequence per GPU: its own scene and noise seeds
    seq = dict(sharding.rank_workload(rank), disparity=wl["disparity"])
    left, right = synth.stereo_pair("s2", H, W, **seq)
    F = synth.fundamental_for(wl["cfg"])

    if wl.get("sequence"):
        return sequence_bench(args, wl, H, W, F, device, rank, world, dist, reduce_device)

    # one context per GPU; --streams S keeps S pairs in flight from this one host thread (S slots; one HIP stream each
    # up to three, four "lane" streams shared by all from five on): submit enqueues a whole pair without host
    # synchronisation, wait blocks on that pair only
    nslots = max(1, args.streams)
    ctx = Context(H, W, device=device, toed_mode=args.toed_mode)
    ctx.set_slots(nslots)
    # the timed pipeline keeps, per candidate pair, what the reference keeps: the final score (max of the four similarities,
    # refine_final_scores, src/Stereo_Matches.cpp:596-600) and the keep flag -- EBVO_PAIR_NO_SIMS; the four similarities are
    # computed either way and are checked after the timed region on a pair submitted with them stored (--store-sims: always)
    params_full = ctx.default_params(F)
    params = ctx.default_params(F)
    if not args.store_sims:
        from edge_based_visual_odometry_amd import _lib as L__
        params.reserved = L__.PAIR_NO_SIMS
    for k in range(nslots):
        ctx.stereo_upload(left, right, slot=k)
    # warm-up: --warmup pairs per slot, and at least MIN_WARM_PAIRS pairs in all (~0.2 s), through the same pipelined submit /
    # wait pattern as the timed region: lane streams, page-locked records and capacity-sized buffers are in their steady state
    # afterwards, and so are the device clocks (they follow the load with a delay: behind the 10 ms that --warmup 5 amounts
    # to, a 20-step timed region read 2730 pairs/s where the same region behind 0.2 s of load reads 2910 and a 300-step
    # region 3040; EBVO_MIN_WARM_PAIRS=0 restores the short form).  The line reports both numbers.
    n_warm = max(max(1, args.warmup) * nslots, int(os.environ.get("EBVO_MIN_WARM_PAIRS", MIN_WARM_PAIRS)))

    def run_pairs(n):
        """n pairs through the pipelined submit / wait pattern; returns (seconds, counts of the last pair)"""
        t_ = time.perf_counter()
        sub_ = done_ = 0
        c_ = None
        while done_ < n:
            while sub_ < n and sub_ - done_ < nslots:
                ctx.stereo_submit(params, slot=sub_ % nslots)
                sub_ += 1
            c_ = ctx.stereo_wait(slot=done_ % nslots)
            done_ += 1
        return time.perf_counter() - t_, c_

    # first the protocol as the command line states it (--warmup W pairs per slot, then K pairs timed; rank-local, no barrier):
    # `value_short_warmup` next to `value`, so that the two protocols can be compared in one record
    n_short = max(1, args.warmup) * nslots
    run_pairs(n_short)
    t_short, counts = run_pairs(args.steps)
    if n_warm > n_short + args.steps:
        _, counts = run_pairs(n_warm - n_short - args.steps)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def tup(c):
        return (c.n_left, c.n_right, c.n_total_left, c.n_total_right, c.n_pairs, c.n_matches)

    # ---- the timed region: submit / wait only ----------------------------------------------------------------
    ctx.profile_enable(False)
    seen = set()
    barrier()
    graphs_before = ctx.graph_launches
    t0 = time.perf_counter()
    submitted = completed = 0
    while submitted < min(nslots, args.steps):
        ctx.stereo_submit(params, slot=submitted % nslots)
        submitted += 1
    last_slot = 0
    while completed < args.steps:                 # EXACTLY args.steps pairs
        k = completed % nslots
        counts = ctx.stereo_wait(slot=k)
        seen.add(tup(counts))
        last_slot = k
        completed += 1
        if submitted < args.steps:
            ctx.stereo_submit(params, slot=k)
            submitted += 1
    barrier()
    dt = time.perf_counter() - t0
    graph_pairs = ctx.graph_launches - graphs_before   # timed pairs whose 18 launches went out as one hipGraphLaunch
    # the sustained rate: 300 pairs more, rank-local, straight after the timed region (the same loop)
    t_sus, _ = run_pairs(300)
    t_sus_full = None
    if not args.store_sims:                          # ... and with the four similarities stored (round 3's pipeline), for comparison
        saved, params = params, params_full
        run_pairs(4 * nslots)                        # (a new flag is a new graph: three submissions per slot)
        t_sus_full, _ = run_pairs(300)
        params = saved
        run_pairs(4 * nslots)
    per_rank = sharding.gather_over_ranks(args.steps / dt, dist, reduce_device)
    dt = sharding.max_over_ranks(dt, dist, reduce_device)
    rank_cpu_counts = [len(c) for c in sharding.gather_int_lists(sorted(os.sched_getaffinity(0)), 1024, dist, reduce_device)]

    # ---- what the timed region produced (every rank checks its own sequence) ---------------------------------
    problems = []
    out = None
    if not args.no_verify:
        if len(seen) != 1:
            problems.append(f"counts changed between steps of the timed region: {sorted(seen)}")
        if not args.store_sims:                     # the same pair once more with all four similarities stored
            ctx.stereo_submit(params_full, slot=last_slot)
            seen.add(tup(ctx.stereo_wait(slot=last_slot)))
            if len(seen) != 1:
                problems.append(f"counts differ between the timed pairs and the pair submitted with the similarities stored: {sorted(seen)}")
        out = ctx.stereo_fetch(counts, slot=last_slot)
        if int(out["keep"].sum()) != counts.n_matches or len(out["col_idx"]) != counts.n_pairs:
            problems.append("fetched arrays disagree with the counts of the run")
        if args.workload == "kitti" and rank == 0:
            got = dict(xyi_left=xyi_hash(out["left"]), xyi_right=xyi_hash(out["right"]), n_left=counts.n_left,
                       n_right=counts.n_right, n_pairs=counts.n_pairs, n_matches=counts.n_matches)
            for key, want in KITTI_KAT.items():
                if got[key] != want:
                    problems.append(f"{key}: {got[key]} != known answer {want}")

    # ---- the same pairs/s with every pair's images arriving from host memory (second key; every rank, same barriers) -----
    ingest = None
    if not args.no_ingest:
        ring = [tuple(np.ascontiguousarray(im) for im in synth.stereo_pair("s2", H, W, scene=seq["scene"], noise_base=seq["noise_base"] + 10 * k,
                                                                            disparity=seq["disparity"])) for k in range(4)]
        for pair in ring:                                   # the caller's frame ring, page-locked once
            for im in pair:
                ctx.host_register(im)
        ctx.set_slots(nslots + 1)                           # nslots pairs in flight + the slot being uploaded a frame ahead
        ingest_loop(ctx, params, ring, nslots, 4 * (nslots + 1))      # untimed: sizes the extra slot, every slot's chain becomes a graph
        barrier()
        t_ing, _ = ingest_loop(ctx, params, ring, nslots, args.steps)
        barrier()
        per_rank_ing = sharding.gather_over_ranks(args.steps / t_ing, dist, reduce_device)
        t_ing = sharding.max_over_ranks(t_ing, dist, reduce_device)
        ingest = {"value_with_h2d": sharding.job_throughput(world, args.steps, t_ing), "per_rank_pairs_per_s_with_h2d": per_rank_ing}
        if rank == 0 and world == 1 and not args.no_transfer_legs:
            n_leg = max(nslots + 1, min(args.steps, 60))
            ingest_host_us = {}
            t_sus_ing, _ = ingest_loop(ctx, params, ring, nslots, 300, host_us=ingest_host_us)
            ingest["ingest_host_us_per_pair"] = ingest_host_us
            ingest_loop(ctx, params, ring, nslots, nslots + 1, "compact")     # untimed: sizes the page-locked staging
            t_c, mb_c = ingest_loop(ctx, params, ring, nslots, n_leg, "compact")
            # ... and with the results PUSHED by the pair's own chain (EBVO_PAIR_PUSH): no copy call, no copy stream
            from edge_based_visual_odometry_amd import _lib as L3
            params_push = ctx.default_params(F)
            params_push.reserved = params.reserved | L3.PAIR_PUSH
            ingest_loop(ctx, params_push, ring, nslots, 4 * (nslots + 1), "push")   # untimed: arenas, graphs of the new flag
            t_p, mb_p = ingest_loop(ctx, params_push, ring, nslots, n_leg, "push")
            t_p300, _ = ingest_loop(ctx, params_push, ring, nslots, 300, "push")
            # ... or PACKED by the chain into device staging and fetched by ONE copy of the copy engine (EBVO_PAIR_PACK)
            params_pack = ctx.default_params(F)
            params_pack.reserved = params.reserved | L3.PAIR_PACK
            ingest_loop(ctx, params_pack, ring, nslots, 4 * (nslots + 1), "compact")
            t_k, mb_k = ingest_loop(ctx, params_pack, ring, nslots, n_leg, "compact")
            t_k300, _ = ingest_loop(ctx, params_pack, ring, nslots, 300, "compact")
            ingest.update(with_h2d_d2h_pack=n_leg / t_k, d2h_bytes_per_pair_pack=mb_k, with_h2d_d2h_pack_sustained_300=300 / t_k300)
            ingest.update(with_h2d_d2h_compact=n_leg / t_c, d2h_bytes_per_pair_compact=mb_c, value_with_h2d_sustained_300=300 / t_sus_ing,
                          with_h2d_d2h_push=n_leg / t_p, d2h_bytes_per_pair_push=mb_p, with_h2d_d2h_push_sustained_300=300 / t_p300)
        ingest["ingest_uploads_by_form"] = ctx.ingest_stats()
        ingest["ingest_note"] = ("value_with_h2d: the timed loop with a NEW pair per step DMA-ed from a page-locked frame ring "
                                 "(ebvo_host_register, ebvo_stereo_upload_async on the context's upload stream, issued one pair ahead of "
                                 "its submission: nslots pairs in flight on nslots + 1 slots), barriers and MAX over ranks as for "
                                 "`value`; with_h2d_d2h_compact adds the compact result fetch ((x, y) of both edge lists, CSR, fp64 "
                                 "best, keep as bits) through page-locked staging; with_h2d_d2h_push: the same arrays written into "
                                 "page-locked host memory by the pair's own chain (EBVO_PAIR_PUSH, ebvo_stereo_pushed_view); with_h2d_d2h_pack: packed by "
                                 "the chain into device staging and fetched by one copy (EBVO_PAIR_PACK)")
        for pair in ring:
            for im in pair:
                ctx.host_unregister(im)
        for k in range(nslots):                                               # restore the resident workload
            ctx.stereo_upload(left, right, slot=k)

    # ---- per-kernel device time, pairs one at a time, HIP events around every kernel (outside the timed region) -
    prof = None
    if rank == 0:
        ctx.profile_reset()
        ctx.profile_enable(True, every=1)
        for _ in range(min(8, max(1, args.steps))):
            ctx.stereo_submit(params, slot=0)
            ctx.stereo_wait(slot=0)
        ctx.profile_enable(False)
        prof = ctx.profile_get()
        # the dominant stage once more with NO other event markers in the stream: markers between all kernels make every
        # kernel start on flushed caches (toed_exact_centre: 98-100 us in the pass above, 91 us here and in a rocprofv3 trace)
        names = list(prof.keys())
        dom_name = max((k for k in names if prof[k][1]), key=lambda k: prof[k][0])
        ctx.debug_set(3, names.index(dom_name) + 1)
        ctx.profile_reset()
        ctx.profile_enable(True, every=1)
        for _ in range(min(8, max(1, args.steps))):
            ctx.stereo_submit(params, slot=0)
            ctx.stereo_wait(slot=0)
        ctx.profile_enable(False)
        prof_dom = ctx.profile_get()[dom_name]
        ctx.debug_set(3, 0)

    # ---- PCIe-inclusive loops (never `value`) -------------------------------------------------------------------
    legs = None
    if rank == 0 and world == 1 and not args.no_transfer_legs:
        pool = [synth.stereo_pair("s2", H, W, scene=seq["scene"], noise_base=seq["noise_base"] + 10 * k,
                                  disparity=seq["disparity"]) for k in range(4)]
        from edge_based_visual_odometry_amd import _lib as L_
        n_leg = max(nslots, min(args.steps, 60))
        frame_loop(ctx, params_full, pool, nslots, nslots, True, L_.FETCH_ALL)     # untimed: touches the pool, sizes the page-locked staging of every slot
        t_up, _ = frame_loop(ctx, params, pool, nslots, n_leg, True, None)
        t_def, mb_def = frame_loop(ctx, params, pool, nslots, n_leg, True, L_.FETCH_DEFAULT)
        t_all, mb_all = frame_loop(ctx, params_full, pool, nslots, n_leg, True, L_.FETCH_ALL)
        t_pg, mb_pg = frame_loop(ctx, params_full, pool, nslots, max(nslots, n_leg // 3), True, "pageable")
        # the one-pass drop-in: get_Stereo_Edge_Pairs as a frame loop calls it through StereoMatcherHIP::stereo_edge_pairs
        cal = synth.CALIB[wl["cfg"]]
        calib = ([cal["K"][0], 0, cal["K"][2], 0, cal["K"][1], cal["K"][3], 0, 0, 1],
                 [cal["K_right"][0], 0, cal["K_right"][2], 0, cal["K_right"][1], cal["K_right"][3], 0, 0, 1], cal["R21"], cal["T21"])
        n_drop = 48                                                                           # (a fixed count: ~0.15 s)
        # untimed: 96 frames (~0.25 s).  Every slot sees every pair of the pool (sizes the chain's buffers) and the loop reaches
        # its steady state: behind a short timed region (--steps 20) a 12-frame warm-up left the 48 timed frames at 290-310
        # frames/s where 96 give the 400-420 every longer run shows (the chain is a sequence of short launches: the device
        # clocks follow the load with a delay)
        dropin_loop(ctx, params, pool, calib, min(nslots, 3), 96)
        t_drop, final_per_pair = dropin_loop(ctx, params, pool, calib, min(nslots, 3), n_drop)
        drop_threads = {str(T): threaded_dropin(T, args, H, W, F, device, pool, calib, 2 * n_drop) for T in host_threads(args)}
        # the drop-in path: what main_VO executes through integration/*.cpp -- host-buffer entry points, results in host
        # arrays, one call after the other (src/Pipeline.cpp:24-29, :109-145): TOED of both images, epipolar lines,
        # candidate search (the three geometric stages in one call), NCC with left patches
        n_b = 6
        split = np.zeros(5)
        for k in range(n_b + 1):                                                  # the first turn is untimed (sizes the buffers)
            bl, br = pool[k % len(pool)]
            tk = [time.perf_counter()]
            eL, _, _, tagL = ctx.toed_resident(bl, 0, want_all=True)
            eR, _, _, tagR = ctx.toed_resident(br, 1, want_all=True)
            tk.append(time.perf_counter())
            lines_b = ctx.epipolar_lines(F, eL)
            tk.append(time.perf_counter())
            ctx.epi_candidates_resident(tagL, tagR, lines_b, staged=True)         # stages 1 + 2 as a list, stage 3 as flags + list
            tk.append(time.perf_counter())
            rp_b, ci_b = ctx.last_final_lists                                     # the caller's lists after the orientation stage
            tk.append(time.perf_counter())
            ctx.ncc_pairs_resident(tagL, tagR, bl, br, rp_b, ci_b, want_left_patches=True, want_sims=False)
            tk.append(time.perf_counter())
            if k:
                split += np.diff(tk)
        split /= n_b
        t_b = float(split.sum())
        boundary_cpp = boundary_bench_cpp(pool[0][0], pool[0][1], H, W, toed_mode=args.toed_mode)  # the same sequence through the C++ adapters
        legs = {"dropin_final_pairs_per_s": n_drop / t_drop, "dropin_final_pairs_per_frame": final_per_pair,
                "dropin_note": "get_Stereo_Edge_Pairs in one pass (StereoMatcherHIP::stereo_edge_pairs_begin / _chain / _end): a new pair "
                               "from host memory per frame, TOED + candidates + NCC, then the enqueue-only chain (SIFT filter, both "
                               "Best-Nearly-Best tests, shift, photometric refinement, clustering, second NCC pass, best per row); "
                               "final pairs + output rows copied back; three frames in flight, two of them in their chains; 48 "
                               "timed frames after 96 untimed ones",
                "dropin_final_pairs_per_s_by_host_threads": drop_threads,
                "dropin_threads_note": "the same loop in T host threads, each with its own context (the library's model: one "
                                       "ebvo_ctx per host thread): the later stages are a host-sequenced chain of short launches, "
                                       "several chains share the device",
                "boundary_pairs_per_s": boundary_cpp["pairs_per_s"] if boundary_cpp else 1.0 / t_b,
                "boundary_cpp": boundary_cpp,
                "boundary_pairs_per_s_python_harness": 1.0 / t_b,
                "boundary_ms": dict(zip(("toed_resident_x2", "epipolar_lines", "epi_candidates_resident_staged", "host_lists", "ncc_pairs_resident_with_left_patches"),
                                        (float(x) * 1e3 for x in split))),
                "boundary_note": "boundary_pairs_per_s: tools/boundary_bench.cpp, the stage-wise sequence as main_VO runs it through "
                                 "include/ebvo/adapters.hpp (ProcessEdges x 2 with the by-value copies of src/Pipeline.cpp:28, "
                                 "CalculateEpipolarLine, one staged candidate search, the lists after the orientation stage, NCC with the "
                                 "left patches); every stage is handed the std::vectors the previous one returned, the adapters recognise "
                                 "them (bit-for-bit comparison) as the edge lists still resident on the device and run the stage there "
                                 "(ebvo_toed_resident / ebvo_epi_candidates_resident / ebvo_ncc_pairs_resident), results are read from "
                                 "page-locked memory; no overlap between calls.  _python_harness / boundary_ms: the same C entry points "
                                 "from numpy (every returned view copied into a fresh array)",
                "with_h2d": n_leg / t_up, "with_h2d_d2h": n_leg / t_def, "d2h_bytes_per_pair": mb_def,
                "with_h2d_d2h_all_scores": n_leg / t_all, "d2h_bytes_per_pair_all_scores": mb_all,
                "with_h2d_d2h_pageable_copies": max(nslots, n_leg // 3) / t_pg, "transfer_leg_pairs": n_leg,
                "transfer_note": "a new pair uploaded from pageable host memory per step (0.93 MB).  d2h = both edge lists, the "
                                 "CSR candidate lists, best score and keep flag per pair through the slot's page-locked staging "
                                 "(ebvo_stereo_fetch_begin / _end, the copy overlaps the other slots' kernels); all_scores adds "
                                 "the four scores per pair; pageable_copies = every array into fresh numpy arrays with "
                                 "synchronous copies (round 1's 343 pairs/s)"}
        for k in range(nslots):                                               # restore the resident workload
            ctx.stereo_upload(left, right, slot=k)

    if rank == 0:
        n_ser = max(1, prof["cand_count"][1])                   # (one launch of the counting pass per pair)
        kernels = {k: {"ms_per_step": v[0] / n_ser, "launches_per_step": v[1] / n_ser} for k, v in prof.items() if v[1]}
        dom = dom_name
        dom_avg_s = prof_dom[0] * 1e-3 / max(1, prof_dom[1])
        dom_avg_all_instrumented_s = prof[dom][0] * 1e-3 / max(1, prof[dom][1])
        alg_bytes = algorithmic_bytes_per_pair(H, W, counts.n_left, counts.n_right, counts.n_pairs)
        achieved_gbs = alg_bytes / dom_avg_s / 1e9
        stats = ctx.toed_stats(0)
        n_cand = stats["left"]["n_candidates"] + stats["right"]["n_candidates"]
        avg_taps = (3 * 361 + 289) / 4.0                       # three 19x19 phases, one 17x17 phase
        fp64_peak, fp64_bound = FP64_VALU_PEAK_NOFMA_TF, "valu_fp64_no_fma"
        if dom == "toed_conv" and args.toed_mode == "strict":
            ops = 2 * H * W * TOED_FLOPS_PER_PX                 # as the reference writes them
            executed = 2 * H * W * 27584.0                      # after forming v*Kcol[q] once per tap
            ops_note = "flops counted as the reference writes them (37,044/px); 27,584/px are executed after CSE"
        elif dom == "toed_exact_centre":
            ops = executed = n_cand * avg_taps * 22.0
            ops_note = "fp64 operations executed: candidates x taps x (4 column products + 9 x (mul, add))"
        elif dom == "toed_exact_mags":
            n_pts = stats["left"]["n_neighbour_points"] + stats["right"]["n_neighbour_points"]
            ops = executed = n_pts * avg_taps * 6.0
            ops_note = ("fp64 operations executed: distinct neighbour grid points (%d, of 4 x %d candidates) x taps x "
                        "(2 column products + 2 x (mul, add)); the id also brackets the decision kernel" % (n_pts, n_cand))
        elif dom == "toed_nms" and args.toed_mode == "hybrid":
            # fused separable screen: per 12 x 30-px block a 32 x 32 row pass with 4 x 19 taps and a 14 x 32 x 4-phase
            # column pass with 2 x (17 | 19) taps, FMA form (the screen is not bound to the reference's arithmetic)
            tiles = 2 * ((H + 11) // 12) * ((W + 29) // 30)
            ops = executed = tiles * (32 * 32 * 76 + 14 * 32 * (2 * 17 + 3 * 2 * 19)) * 2.0
            ops_note = "flops of the separable FP32 screen (FMA = 2), relaxed NMS not counted"
            fp64_peak, fp64_bound = 4 * FP64_VALU_PEAK_NOFMA_TF, "valu_fp32_fma"   # 157 TFLOP/s: FP32 vector FMA (MI355X_MICROARCH.md)
        else:
            ops, executed, ops_note = None, None, "not an fp64-ALU kernel"
        result = {
            "metric": "stereo pairs/sec (TOED+NCC match) on KITTI 1241\u00d7376; achieved HBM GB/s",   # BASELINE.json's string
            "value": sharding.job_throughput(world, args.steps, dt),
            "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_pairs_run": max(n_warm, n_short + args.steps), "warmup_overridden": n_warm > n_short,
            "value_short_warmup": args.steps / t_short, "value_sustained_300": 300 / t_sus,
            "value_sustained_300_sims_stored": None if t_sus_full is None else 300 / t_sus_full,
            "warmup_note": f"value: {args.steps} pairs timed after {max(n_warm, n_short + args.steps)} untimed ones (the device clocks follow "
                           f"the load with a delay); value_short_warmup: the same {args.steps} pairs timed after --warmup x slots = "
                           f"{n_short} pairs only (this rank, no barrier); value_sustained_300: 300 pairs straight after the timed region",
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "per_rank_pairs_per_s": per_rank,
            **sharding.rank_balance(per_rank, world, sharding.job_throughput(world, args.steps, dt)),
            "rank_cpus": rank_cpu_counts,
            "config": {"workload": wl["label"], "shape": f"{W}x{H}",
                       "toed_mode": args.toed_mode,
                       "edges_left": counts.n_left, "edges_right": counts.n_right,
                       "toed_candidates": n_cand if args.toed_mode == "hybrid" else None,
                       "candidate_pairs": counts.n_pairs, "ncc_matches": counts.n_matches,
                       "sims_stored": bool(args.store_sims),
                       "pairs_in_flight_per_gpu": nslots,
                       "streams_per_gpu": nslots if nslots < 4 else min(4, nslots - 1),
                       "pairs_submitted_as_hipgraph": graph_pairs, "toed_strict_fallbacks": ctx.toed_fallbacks,
                       "parallelism": f"{world} independent sequence(s), one per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, args.toed_mode),
                         "traffic_source": f"profiles/kernel_pmc_{args.toed_mode}.json (committed rocprofv3 PMC passes, "
                                           "FETCH_SIZE x 2 + WRITE_SIZE per launch; not measured in this run)",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_avg_s * 1e3,
                         "avg_launch_ms_every_stage_instrumented": dom_avg_all_instrumented_s * 1e3,
                         "note": "achieved = the whole pair's algorithmic bytes (SURVEY.md 8(d)) / the dominant kernel's "
                                 "launch duration: the contract's formula, not a bandwidth this kernel moves.  The path is "
                                 "FP64-VALU-bound, not HBM-bound; see roofline_fp64.  Durations: HIP events, pairs one at a "
                                 "time, after the timed region; avg_launch_ms: the dominant kernel alone carries events (stamped by "
                                 "its own dispatch, hipExtLaunchKernelGGL), as in a rocprofv3 kernel trace; with event markers "
                                 "between ALL kernels (the pass behind `kernels`) every kernel starts on flushed caches and "
                                 "reads longer (avg_launch_ms_every_stage_instrumented)."},
            "kernels": kernels,
            "kernel_groups": KERNEL_GROUPS[args.toed_mode],
        }
        if args.workload != "kitti":
            result["metric"] = f"stereo pairs/sec (TOED+NCC match) on {wl['cfg']} {W}x{H}; achieved HBM GB/s"
        if ops is not None:
            tf = executed / dom_avg_s / 1e12               # operations the kernel EXECUTES (strict mode: after CSE)
            result["roofline_fp64"] = {"bound": fp64_bound, "kernel": dom, "achieved": tf,
                                       "peak": fp64_peak, "unit": "TFLOP/s", "frac": tf / fp64_peak,
                                       "ops_per_launch": ops, "executed_ops_per_launch": executed,
                                       "executed_frac": executed / dom_avg_s / 1e12 / fp64_peak,
                                       "frac_counting_reference_ops": ops / dom_avg_s / 1e12 / fp64_peak,
                                       "peak_sustained_measured": fp64_peak * FP64_VALU_SUSTAINED_FRAC,
                                       "frac_of_sustained": executed / dom_avg_s / 1e12 / (fp64_peak * FP64_VALU_SUSTAINED_FRAC),
                                       "note": ops_note + "; peak = 78.6 TFLOP/s vendor FP64 vector (FMA), halved where mul "
                                                          "and add must stay separate operations"}
        if ingest is not None:
            result.update(ingest)
        if legs is not None:
            result.update(legs)
        if world > 1:
            result["cpu_baseline"] = None
            result["cpu_baseline_note"] = "N > 1: the CPU baseline is timed on rank 0 at N = 1 only"
        if world == 1 and not args.no_cpu_baseline:
            rec, chk = cpu_baseline(left, right, F)
            result["cpu_baseline"] = rec
            result["gpu_over_cpu"] = result["value"] / rec["value"]
            if out is not None:
                bad = check_against_oracle(out, chk)
                if bad:
                    problems.append(bad)
                result["verified_against_cpu_baseline"] = bad is None
                # ... and against the baseline's results in the reference's OWN arithmetic (glibc), north_star's bar
                rep = reference_arithmetic_report(ctx, out, chk["libm"], F)
                result["reference_arithmetic"] = rep
                from tests.reference_arithmetic import BOOLEANS
                for key in BOOLEANS:
                    result[key] = rep[key]
                    if not rep[key]:
                        problems.append(f"{key} is false: {rep['flips'][:4]}")
        if not args.no_verify:
            result["verified"] = not problems
            if problems:
                result["verification_failures"] = problems
        print(json.dumps(result))
    elif problems:
        print(f"[rank {rank}] verification failed: {problems}", file=sys.stderr)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if problems:
        sys.exit(3)


if __name__ == "__main__":
    main()
