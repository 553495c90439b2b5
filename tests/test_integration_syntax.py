"""The reference-side translation units (integration/*.cpp) against the reference's REAL headers: every replacement body
must still match the class declarations of include/Stereo_Matches.h:58-104, include/Temporal_Matches.h:66 and
include/toed/cpu_toed.hpp:70-115, and use the reference's containers (Stereo_Edge_Pairs, EdgeCluster, final_stereo_edge_pair
...) as they are declared.  `g++ -fsyntax-only` with declaration-only stand-ins for OpenCV / Eigen / yaml-cpp (tests/stubs):
a check of syntax and signatures, never parity evidence.  Runs only where /root/reference exists (the build container);
nothing of the reference is committed or shipped."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/include"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not here")

UNITS = ["hip_toed.cpp", "stereo_matches_hip.cpp"]


def syntax_check(path):
    return subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-fopenmp", "-I", os.path.join(ROOT, "tests", "stubs"), "-I", REF,
                           "-I", os.path.join(ROOT, "include"), path], capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("unit", UNITS)
def test_translation_unit_parses_against_the_reference_headers(unit):
    out = syntax_check(os.path.join(ROOT, "integration", unit))
    assert out.returncode == 0, out.stderr[-4000:]


def test_the_check_notices_a_signature_that_drifted(tmp_path):
    """negative control: the same unit with one parameter type changed must NOT pass"""
    src = open(os.path.join(ROOT, "integration", "stereo_matches_hip.cpp")).read()
    changed, n = re.subn(r"void Stereo_Matches::apply_NCC_Filtering\(Stereo_Edge_Pairs &p, const std::string &, size_t, bool is_left\)",
                         "void Stereo_Matches::apply_NCC_Filtering(Stereo_Edge_Pairs &p, const std::string &, int, bool is_left)", src)
    assert n == 1
    path = tmp_path / "drifted.cpp"
    path.write_text(changed)
    out = syntax_check(str(path))
    assert out.returncode != 0 and "apply_NCC_Filtering" in out.stderr
    # ... and a member the reference's record does not have
    changed2 = src.replace("fe.right_edge.orientation = re.theta;", "fe.right_edge.orientation = re.theta;\n        fe.no_such_member = 1;")
    assert changed2 != src
    path.write_text(changed2)
    out = syntax_check(str(path))
    assert out.returncode != 0 and "no_such_member" in out.stderr
