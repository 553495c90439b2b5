import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_based_visual_odometry_amd.api import Context
with Context(64, 64) as c:
    for _ in range(3):
        a, b = c.fp64_peak(20)
        print(f"fp64 VALU: mul+add (no FMA) {a:.2f} TFLOP/s   fma {b:.2f} TFLOP/s")
