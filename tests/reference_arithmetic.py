"""GPU results against the reference's OWN elementary functions (glibc sin / cos / atan2), end to end.

Every other parity test compares the HIP path with the oracle's PORTABLE mode, where theta, sin and cos come from the
one header both sides share (csrc/ebvo_math.h): a self-comparison for those three functions.  Here the oracle runs in
LIBM mode -- `atan2` of src/toed/cpu_toed.cpp:229, `sin` / `cos` of src/utility.cpp:84-87, :146-151 are the C library's,
as in the reference's build -- and the GPU is held to north_star's bar against THAT:

  * (x, y, index) of every edge bit-equal (no elementary function is involved: src/toed/cpu_toed.cpp:505-510, :538-562);
  * |theta_gpu - theta_libm| <= 1 ulp (both are faithful roundings of the same real number);
  * the candidate lists after each of the three geometric stages identical (src/Stereo_Matches.cpp:91-109, :534-553,
    :863-915; the orientation gate :887-901 is the only one that reads theta);
  * `keep` of the NCC filter identical (:592-597) and max |sims_gpu - sims_libm| <= 1e-5 (src/utility.cpp:163-180).

A decision that flips is REPORTED with its margin (how far the deciding quantity sits from its threshold), never hidden:
`flips` lists them.  Checker-side code: only tests/ and bench.py's verification leg (after the timed region) import this.
"""
from __future__ import annotations

import os

import numpy as np

from tests import oracle as orc

SIM_TOL = 1e-5          # north_star: "NCC scores within 1e-5"


def _ulp_diff(a, b):
    """|a - b| in units of the spacing at b (finite inputs)."""
    return np.abs(a - b) / np.spacing(np.abs(b))


def _rows_of(row_ptr):
    return np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int64), np.diff(row_ptr.astype(np.int64)))


def _keys(row_ptr, col_idx):
    return (_rows_of(row_ptr) << 32) | col_idx.astype(np.int64)


def _csr_from_flags(row_ptr, col_idx, flags, n_rows):
    rows = _rows_of(row_ptr)
    sel = flags.astype(bool)
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows[sel], minlength=n_rows))]).astype(np.int32)
    return rp, col_idx[sel]


def _orientation_margin_deg(tl, tr):
    """distance of the orientation gate's quantity from its 10-degree threshold (src/Stereo_Matches.cpp:887-901)"""
    d = np.abs(np.degrees(tl - tr))
    d = np.where(d > 180.0, 360.0 - d, d)
    return np.minimum(np.abs(d - 10.0), np.abs(np.abs(d - 180.0) - 10.0))


def oracle_libm_stages(toed_left, toed_right, ncc_left, ncc_right, F, cores=0, with_epipolar_stage=True):
    """The reference path in LIBM arithmetic, stage after stage as get_Stereo_Edge_Pairs runs it
    (src/Stereo_Matches.cpp:1374-1427).  toed_* are the images the detector sees (undistorted where the configuration
    undistorts, src/Pipeline.cpp:93-97), ncc_* the raw ones (src/Stereo_Matches.cpp:562-563)."""
    L = orc.toed(toed_left, math_mode=orc.LIBM, nthreads=cores)["edges"]
    R = orc.toed(toed_right, math_mode=orc.LIBM, nthreads=cores)["edges"]
    lines = orc.epipolar_lines(F, L)
    out = dict(left=L, right=R, lines=lines)
    if with_epipolar_stage:
        rp1, ci1 = orc.epi_candidates(L, R, lines, stage_mask=orc.STAGE_EPIPOLAR, nthreads=cores)
        k2 = orc.filter_pairs(L, R, rp1, ci1, stage_mask=orc.STAGE_DISPARITY, nthreads=cores)
        rp2, ci2 = _csr_from_flags(rp1, ci1, k2, len(L))
        out["stage1"] = (rp1, ci1)
    else:
        rp2, ci2 = orc.epi_candidates(L, R, lines, stage_mask=orc.STAGE_EPIPOLAR | orc.STAGE_DISPARITY, nthreads=cores)
    k3 = orc.filter_pairs(L, R, rp2, ci2, stage_mask=orc.STAGE_ORIENTATION, nthreads=cores)
    rp3, ci3 = _csr_from_flags(rp2, ci2, k3, len(L))
    sims, best, keep, _ = orc.ncc_pairs(ncc_left, ncc_right, L, R[ci3], rp3, math_mode=orc.LIBM, nthreads=cores)
    out.update(stage2=(rp2, ci2), stage3=(rp3, ci3), sims=sims, best=best, keep=keep)
    return out


def compare(ref, gpu, ncc_thr=0.6, max_flips_listed=16):
    """ref: oracle_libm_stages(...); gpu: dict(left, right, stage2=(rp, ci), stage3=(rp, ci), sims, best, keep[, stage1]).
    Returns the report: the four booleans of the bench line + sizes, maxima and every flip with its margin."""
    rep = {"flips": []}
    ids = True
    max_ulp = 0.0
    n_theta_diff = 0
    for side in ("left", "right"):
        a, b = gpu[side], ref[side]
        same = len(a) == len(b) and all((a[f].view(np.uint64) == b[f].view(np.uint64)).all() for f in ("x", "y")) and \
            bool((a["index"] == b["index"]).all())
        ids = ids and same
        if same and len(a):
            u = _ulp_diff(a["theta"], b["theta"])
            max_ulp = max(max_ulp, float(u.max()))
            n_theta_diff += int((a["theta"].view(np.uint64) != b["theta"].view(np.uint64)).sum())
    rep["ids_equal_reference_arithmetic"] = bool(ids)
    rep["theta_max_ulp_vs_libm"] = max_ulp
    rep["theta_within_1ulp_of_libm"] = bool(ids and max_ulp <= 1.0)
    rep["theta_differing_from_libm"] = n_theta_diff
    rep["edges"] = [int(len(ref["left"])), int(len(ref["right"]))]
    if not ids:
        rep.update(stages_equal_reference_arithmetic=False, keep_equal_reference_arithmetic=False,
                   sims_max_abs_diff_vs_libm=None, sims_within_1e5_of_libm=False)
        return rep

    stages_equal = True
    for st in ("stage1", "stage2", "stage3"):
        if st not in ref or st not in gpu:
            continue
        (rp_r, ci_r), (rp_g, ci_g) = ref[st], gpu[st]
        eq = np.array_equal(rp_r, rp_g) and np.array_equal(ci_r, ci_g)
        rep[f"{st}_pairs"] = int(len(ci_r))
        rep[f"{st}_equal"] = bool(eq)
        stages_equal = stages_equal and eq
        if not eq:
            kr, kg = _keys(rp_r, ci_r), _keys(rp_g, ci_g)
            for who, only in (("reference_only", np.setdiff1d(kr, kg)), ("gpu_only", np.setdiff1d(kg, kr))):
                for key in only[:max_flips_listed]:
                    i, j = int(key >> 32), int(key & 0xFFFFFFFF)
                    m = float(_orientation_margin_deg(ref["left"]["theta"][i:i + 1], ref["right"]["theta"][j:j + 1])[0])
                    rep["flips"].append(dict(stage=st, pair=[i, j], side=who,
                                             orientation_margin_deg=m if st == "stage3" else None))
                rep[f"{st}_{who}"] = int(len(only))
    rep["stages_equal_reference_arithmetic"] = bool(stages_equal)

    # NCC: compare on the pairs both sides list after the orientation stage (all of them when the stages are equal)
    kr, kg = _keys(*ref["stage3"]), _keys(*gpu["stage3"])
    if stages_equal:
        ir = ig = np.arange(len(kr))
    else:
        _, ir, ig = np.intersect1d(kr, kg, assume_unique=True, return_indices=True)
    sr, sg = ref["sims"][ir], gpu["sims"][ig]
    both_nan = np.isnan(sr) & np.isnan(sg)
    nan_mismatch = int((np.isnan(sr) != np.isnan(sg)).sum())
    d = np.where(both_nan, 0.0, np.abs(sr - sg))
    d = np.where(np.isnan(d), np.inf, d)
    rep["sims_compared"] = int(sr.size)
    rep["sims_nan_mismatches"] = nan_mismatch
    rep["sims_max_abs_diff_vs_libm"] = float(d.max()) if d.size else 0.0
    rep["sims_bit_equal_fraction"] = float((sr.view(np.uint64) == sg.view(np.uint64)).mean()) if d.size else 1.0
    rep["sims_within_1e5_of_libm"] = bool(nan_mismatch == 0 and (d.size == 0 or d.max() <= SIM_TOL))
    kk_r, kk_g = ref["keep"][ir], gpu["keep"][ig]
    flips = np.flatnonzero(kk_r != kk_g)
    rep["keep_flips"] = int(len(flips))
    rep["ncc_matches"] = int(ref["keep"].sum())
    for f in flips[:max_flips_listed]:
        key = int(kr[ir][f])
        rep["flips"].append(dict(stage="ncc_keep", pair=[key >> 32, key & 0xFFFFFFFF],
                                 best_reference=float(ref["best"][ir][f]), best_gpu=float(gpu["best"][ig][f]),
                                 margin=float(abs(ref["best"][ir][f] - ncc_thr))))
    rep["keep_equal_reference_arithmetic"] = bool(stages_equal and len(flips) == 0)
    return rep


def gpu_stages(ctx, toed_left, toed_right, ncc_left, ncc_right, F, with_epipolar_stage=True):
    """The same stages through the library's host-buffer entry points (ebvo_toed_pair, ebvo_epipolar_lines,
    ebvo_epi_candidates with the reference's stage masks, ebvo_ncc_pairs)."""
    L, R, _ = ctx.toed_pair(toed_left, toed_right)
    lines = ctx.epipolar_lines(F, L)
    out = dict(left=L, right=R)
    if with_epipolar_stage:
        out["stage1"] = ctx.epi_candidates(L, R, lines, stage_mask=orc.STAGE_EPIPOLAR)
    out["stage2"] = ctx.epi_candidates(L, R, lines, stage_mask=orc.STAGE_EPIPOLAR | orc.STAGE_DISPARITY)
    rp3, ci3 = ctx.epi_candidates(L, R, lines, stage_mask=orc.STAGE_ALL)
    out["stage3"] = (rp3, ci3)
    sims, best, keep, _ = ctx.ncc_pairs(ncc_left, ncc_right, L, R[ci3], rp3)
    out.update(sims=sims, best=best, keep=keep)
    return out


def default_cores():
    return int(os.environ.get("EBVO_CPU_THREADS", min(16, len(os.sched_getaffinity(0)))))


BOOLEANS = ("ids_equal_reference_arithmetic", "theta_within_1ulp_of_libm", "stages_equal_reference_arithmetic",
            "keep_equal_reference_arithmetic", "sims_within_1e5_of_libm")
