#!/usr/bin/env python3
"""Build tests/golden/ncc_debug_frame1_edge8.json from the reference's own NCC fixture.

Source (data files only, read as data): /root/reference/test/ncc_debug_frame1_edge8/
  {prev,gt,cand1..5}_patch_{plus,minus}.png  -- 7x7 patches, upscaled x20 (nearest) and min-max
                                                normalised to 0..255 (test/debug_ncc_patches.m:95-96,690-707)
  patch_statistics.txt                        -- the four NCC values per candidate, 4 decimals
The PNGs are reduced back to 7x7 by sampling the centre of each 20x20 cell.  NCC is invariant to
the per-patch affine normalisation, so the statistics are reproducible from the PNGs to ~2e-3
(8-bit quantisation).  Run in the build container (needs /root/reference and PIL); the JSON is
committed so the tests need neither.
"""
import json
import os
import re

import numpy as np
from PIL import Image

SRC = "/root/reference/test/ncc_debug_frame1_edge8"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ncc_debug_frame1_edge8.json")


def patch(name):
    im = np.array(Image.open(os.path.join(SRC, name)).convert("L"))
    assert im.shape == (140, 140), im.shape
    return im[10::20, 10::20].astype(int).tolist()


names = ["prev", "gt"] + [f"cand{k}" for k in range(1, 6)]
patches = {n: {"plus": patch(f"{n}_patch_plus.png"), "minus": patch(f"{n}_patch_minus.png")} for n in names}

txt = open(os.path.join(SRC, "patch_statistics.txt")).read()
blocks = re.split(r"\n(?=NCC Scores between|Candidate \d+:)", txt)
expected = {}
for b in blocks:
    m = re.search(r"Plus-Plus: ([-\d.]+)\s+Minus-Minus: ([-\d.]+)\s+Plus-Minus: ([-\d.]+)\s+Minus-Plus: ([-\d.]+)\s+Max: ([-\d.]+)", b)
    if not m:
        continue
    vals = [float(v) for v in m.groups()]
    if b.startswith("NCC Scores between"):
        expected["gt"] = vals
    else:
        k = int(re.match(r"Candidate (\d+):", b).group(1))
        expected[f"cand{k}"] = vals
assert len(expected) == 6, expected.keys()
json.dump({"source": "test/ncc_debug_frame1_edge8 (reference repository)",
           "order": ["plus-plus", "minus-minus", "plus-minus", "minus-plus", "max"],
           "tolerance": 2e-3, "patches": patches, "expected_vs_prev": expected}, open(OUT, "w"), indent=1)
print("wrote", OUT)
