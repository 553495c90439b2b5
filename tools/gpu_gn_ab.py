"""Run ON the GPU box: A/B of the stereo refinement's launch layouts on the one-pass chain (KITTI-size pair, with the SIFT
stages) and on the EuRoC-size frame: persistent eight-lanes launch built for 2 / 3 waves per SIMD, and a launch per
iteration (the form before).  Wall time of ebvo_stereo_finalize, best of 5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context

for cfg, disp in (("kitti", 12), ("euroc", 9)):
    h, w = synth.SHAPES[cfg]
    F = synth.fundamental_for(cfg)
    cal = synth.CALIB[cfg]
    calib = ([cal["K"][0], 0, cal["K"][2], 0, cal["K"][1], cal["K"][3], 0, 0, 1],
             [cal["K_right"][0], 0, cal["K_right"][2], 0, cal["K_right"][1], cal["K_right"][3], 0, 0, 1], cal["R21"], cal["T21"])
    l, r = synth.stereo_pair("s2", h, w, disparity=disp)
    with Context(h, w) as c:
        c.stereo_upload(l, r)
        c.stereo_run(c.default_params(F))
        c.stereo_finalize(calib, use_sift=True)
        variants = [("eight-lanes layout as a launch per iteration (round 2's form)", ((7, 1), (8, 2), (9, 0), (5, 49152))),
                    ("persistent launch (default: <= 1024 blocks, below 65536)", ((7, 0), (8, 2), (9, 0), (5, 0))),
                    ("persistent, built for 3 waves / SIMD", ((7, 0), (8, 3), (9, 0), (5, 0)))]
        for blocks in (512, 1024, 8192):
            variants.append((f"persistent, <= {blocks} blocks", ((7, 0), (8, 2), (9, blocks), (5, 0))))
        for below in (32768, 131072, 262144):
            variants.append((f"persistent, eight-lanes layout below {below}", ((7, 0), (8, 2), (9, 0), (5, below))))
        for name, keys in variants:
            for k, v in keys:
                c.debug_set(k, v)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                counts, _ = c.stereo_finalize(calib, use_sift=True)
                best = min(best, time.perf_counter() - t0)
            c.profile_reset(); c.profile_enable(True)
            c.stereo_finalize(calib, use_sift=True)
            c.profile_enable(False)
            gn = c.profile_get()["gn_refine"][0]
            print(f"{cfg}: {name:58s} finalize {best * 1e3:6.3f} ms, refinement stage {gn:6.3f} ms  (pairs refined: {counts['n_bnb']})")
