"""The reference's one numeric fixture -- test/ncc_debug_frame1_edge8 (12 patch PNGs + patch_statistics.txt:20-80),
committed as tests/golden/ncc_debug_frame1_edge8.json -- applied to the HIP path DIRECTLY, not through the oracle:

* ebvo_ncc_patches (the C entry a Utility::get_patch_similarity / MatlabNCCComputer::computeNCC caller binds,
  src/utility.cpp:163-180, include/MatlabNCCComputer.h:41) through ctypes,
* ebvo::patch_similarity (include/ebvo/adapters.hpp) from a C++ program built with plain g++,
* ebvo_ncc_quads (Temporal_Matches::apply_NCC_filtering_quads' scorer, src/Temporal_Matches.cpp:440-451: the
  maximum of the four combinations).

Order of the four scores: PP, MM, PM, MP, then their maximum (src/Stereo_Matches.cpp:592-596).  Tolerance 2e-3: the
fixture's patches are 8-bit PNGs of min-max normalised data and its scores are printed to four decimals."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from edge_based_visual_odometry_amd import _lib
from tests.util import GOLDEN, assert_bit_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "ncc_fixture_demo.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "ncc_fixture_demo")


def fixture_pairs():
    """-> names, A (n x 49), B (n x 49), expected (n,) in the order PP, MM, PM, MP per candidate, and the per-candidate max"""
    d = json.load(open(os.path.join(GOLDEN, "ncc_debug_frame1_edge8.json")))
    prev = d["patches"]["prev"]
    A, B, exp, names, exp_max = [], [], [], [], []
    for name, e in d["expected_vs_prev"].items():
        c = d["patches"][name]
        for a, b in ((prev["plus"], c["plus"]), (prev["minus"], c["minus"]), (prev["plus"], c["minus"]),
                     (prev["minus"], c["plus"])):
            A.append(np.asarray(a, dtype=np.float32).reshape(49))
            B.append(np.asarray(b, dtype=np.float32).reshape(49))
        exp.extend(e[:4])
        exp_max.append(e[4])
        names.append(name)
    return names, np.stack(A), np.stack(B), np.array(exp), np.array(exp_max), d["tolerance"]


def build_demo():
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                           "-L", libdir, "-lebvo_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])


def test_fixture_demo_builds_with_plain_gxx():
    build_demo()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_gpu_ncc_patches_reproduces_reference_fixture(ctx):
    names, A, B, exp, exp_max, tol = fixture_pairs()
    assert len(names) == 6 and tol == 2e-3
    got = ctx.ncc_patches(A, B)
    err = np.abs(got - exp)
    assert err.max() <= tol, (names[int(err.argmax()) // 4], got.reshape(-1, 4), exp.reshape(-1, 4))
    # the candidate's final score: best = pp; if (best < nn) ...  (src/Stereo_Matches.cpp:596)
    assert np.abs(got.reshape(-1, 4).max(axis=1) - exp_max).max() <= tol


@pytest.mark.gpu
def test_gpu_ncc_quads_reproduces_reference_fixture(ctx):
    """ebvo_ncc_quads takes (plus, minus) patch pairs of both views and returns max-of-four per side: feeding the fixture's
    `prev` as the keyframe mate and each candidate as the current-frame mate must give the fixture's max column."""
    names, _, _, _, exp_max, tol = fixture_pairs()
    d = json.load(open(os.path.join(GOLDEN, "ncc_debug_frame1_edge8.json")))["patches"]
    kf = np.stack([np.stack([np.asarray(d["prev"]["plus"], np.float32).reshape(49),
                             np.asarray(d["prev"]["minus"], np.float32).reshape(49)])] * len(names))
    cf = np.stack([np.stack([np.asarray(d[n]["plus"], np.float32).reshape(49),
                             np.asarray(d[n]["minus"], np.float32).reshape(49)]) for n in names])
    sl, sr, keep = ctx.ncc_quads(kf, kf, cf, cf, thr=0.8)
    assert np.abs(sl - exp_max).max() <= tol and np.abs(sr - exp_max).max() <= tol
    # every expected max is >= 0.8472 > 0.8 + tol: the keep flag of src/Temporal_Matches.cpp:452 is decided by the fixture
    assert keep.all()


@pytest.mark.gpu
def test_cpp_patch_similarity_reproduces_reference_fixture(tmp_path):
    build_demo()
    names, A, B, exp, exp_max, tol = fixture_pairs()
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<i", len(A)))
        for a, b in zip(A, B):
            f.write(a.tobytes())
            f.write(b.tobytes())
    subprocess.check_call([EXE, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    out = np.frombuffer((tmp_path / "out.bin").read_bytes(), dtype=np.float64)
    single, batch = out[: len(A)], out[len(A):]
    assert_bit_equal(single, batch, "one call per pair vs one batched call")
    assert np.abs(single - exp).max() <= tol, (single.reshape(-1, 4), exp.reshape(-1, 4))
    assert np.abs(single.reshape(-1, 4).max(axis=1) - exp_max).max() <= tol
