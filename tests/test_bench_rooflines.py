"""The roofline objects bench.py reads from the COMMITTED rocprofv3 passes (profiles/): they must parse, and say what they are."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module_for_tests", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_sift_roofline_reads_the_committed_counter_pass():
    r = _bench().sift_roofline()
    assert r is not None, "profiles/r04_chain_pmc_sq_chain.txt holds no SIFT descriptor kernel"
    assert r["bound"] == "valu_issue" and r["unit"] == "G wave-instructions/s"
    assert r["kernel"].startswith("sift_desc")
    assert abs(r["peak"] - 614.4) < 1e-9                       # 1,024 SIMDs x 2.4 GHz / 4 cycles
    assert 0.2 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = waves x instructions per wave / launch duration
    assert abs(r["achieved"] - r["waves"] * r["valu_instructions_per_wave"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6
    assert "not measured in this run" in r["source"]


def test_committed_hbm_counters_of_the_dominant_kernel_parse():
    b = _bench()
    t = b.pmc_traffic("toed_exact_centre", "hybrid")
    assert t is None or t > 0
