"""Seeded synthetic stereo images at the dataset shapes.

No dataset (KITTI / EuRoC / ETH3D) ships with the reference or exists on the GPU box, so every
test and the benchmark use the two integer-only generators defined with known-answer hashes in
SURVEY.md section 8(c):

* S1 -- two-plateau checker pattern + uniform noise (cheap smoke test),
* S2 -- multi-octave integer value noise (textured, isotropic edge orientations; the primary
  workload: ~126 k third-order edges per 1241x376 image).

Both draw their noise from ``std::mt19937`` (one draw per pixel, row-major); numpy's legacy
MT19937 seeding reproduces that stream exactly.  ``img_fnv`` validates a generated image against
the hashes recorded in the survey before it is used.
"""
from __future__ import annotations

import numpy as np

SHAPES = {
    "kitti": (376, 1241),   # config/kitti.yaml:13
    "euroc": (480, 752),    # config/euroc.yaml
    "eth3d": (489, 942),    # config/eth3d_delivery_area.yaml:11
    "tiny": (48, 64),
}


def _mt19937_raw(seed: int, n: int) -> np.ndarray:
    bg = np.random.MT19937()
    bg._legacy_seeding(int(seed))  # == std::mt19937(seed)
    return bg.random_raw(n).astype(np.uint32)


def s1_image(h: int, w: int, seed: int = 1, shift: int = 0) -> np.ndarray:
    """S1: v(i,j) = ((i/23 + (j+shift)/31) % 2)*120 + 40 + (mt() % 9)."""
    i = np.arange(h, dtype=np.int64)[:, None]
    j = np.arange(w, dtype=np.int64)[None, :]
    noise = (_mt19937_raw(seed, h * w) % 9).astype(np.int64).reshape(h, w)
    v = ((i // 23 + (j + shift) // 31) % 2) * 120 + 40 + noise
    return v.astype(np.uint8)


def _lat(o: int, y: np.ndarray, x: np.ndarray, s: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        hsh = (x.astype(np.uint32) * np.uint32(73856093)) ^ (y.astype(np.uint32) * np.uint32(19349663))
        hsh = hsh ^ np.uint32((o * 83492791) & 0xFFFFFFFF) ^ np.uint32((s * 2654435761) & 0xFFFFFFFF)
        hsh = hsh ^ (hsh >> np.uint32(13))
        hsh = hsh * np.uint32(0x5BD1E995)
        hsh = hsh ^ (hsh >> np.uint32(15))
    return (hsh & np.uint32(255)).astype(np.int64)


def _octave(o: int, i: np.ndarray, j: np.ndarray, s: int) -> np.ndarray:
    yi, xi, fy, fx = i // o, j // o, i % o, j % o
    a = _lat(o, yi, xi, s)
    b = _lat(o, yi, xi + 1, s)
    c = _lat(o, yi + 1, xi, s)
    d = _lat(o, yi + 1, xi + 1, s)
    return ((a * (o - fx) + b * fx) * (o - fy) + (c * (o - fx) + d * fx) * fy) // (o * o)


def s2_image(h: int, w: int, scene: int = 7, noise_seed: int = 1, shift: int = 0) -> np.ndarray:
    """S2: clamp((V4 + 2*V8 + 2*V16 + 3*V32)/8 + (mt()%3) - 1, 0, 255) sampled at (i, j+shift)."""
    i = np.broadcast_to(np.arange(h, dtype=np.int64)[:, None], (h, w))
    j = np.broadcast_to(np.arange(w, dtype=np.int64)[None, :] + shift, (h, w))
    v = (_octave(4, i, j, scene) + 2 * _octave(8, i, j, scene) + 2 * _octave(16, i, j, scene)
         + 3 * _octave(32, i, j, scene)) // 8
    noise = (_mt19937_raw(noise_seed, h * w) % 3).astype(np.int64).reshape(h, w) - 1
    return np.clip(v + noise, 0, 255).astype(np.uint8)


def stereo_pair(kind: str, h: int, w: int, scene: int = 7, noise_base: int = 0, disparity: int = 12):
    """Left = G(noise 1, shift 0), right = G(noise 2, shift disparity): the scene appears
    ``disparity`` px further left in the right image (x_R = x_L - d, src/Stereo_Matches.cpp:159)."""
    if kind == "s1":
        return s1_image(h, w, noise_base + 1, 0), s1_image(h, w, noise_base + 2, disparity)
    if kind == "s2":
        return (s2_image(h, w, scene, noise_base + 1, 0), s2_image(h, w, scene, noise_base + 2, disparity))
    raise ValueError(kind)


def img_fnv(img: np.ndarray) -> str:
    """FNV-1a-64 over the u8 pixels in row-major order, as 16 hex digits."""
    h = 1469598103934665603
    prime = 1099511628211
    mask = (1 << 64) - 1
    for b in np.ascontiguousarray(img, dtype=np.uint8).tobytes():
        h = ((h ^ b) * prime) & mask
    return f"{h:016x}"


# Calibrations of the reference configs (config/kitti.yaml:13-28 etc.): fx, fy, cx, cy, R21, T21.
CALIB = {
    # config/kitti.yaml:13-28 (rectified)
    "kitti": dict(K=(718.856, 718.856, 607.1928, 185.2157), K_right=(718.856, 718.856, 607.1928, 185.2157),
                  R21=((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)), T21=(0.54, 0.0, 0.0)),
    # config/euroc.yaml:10-28 (NOT rectified: slanted epipolar lines)
    "euroc": dict(K=(458.654, 457.296, 367.215, 248.375), K_right=(457.587, 456.134, 379.999, 255.238),
                  R21=((0.999997256477450, 0.002312067192420, 0.000376008102351),
                       (-0.002317135723285, 0.999898048506528, 0.014089835846697),
                       (-0.000343393120589, -0.014090668452670, 0.999900662638179)),
                  T21=(-0.110073808127139, 0.000399121547014534, -0.000853702503351098),
                  # config/euroc.yaml:13, :18: k1 k2 p1 p2
                  dist=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05),
                  dist_right=(-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05)),
    # config/eth3d_delivery_area.yaml:10-28
    "eth3d": dict(K=(541.764, 541.764, 553.869, 232.396), K_right=(541.764, 541.764, 553.869, 232.396),
                  R21=((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)),
                  T21=(-0.059891300000001, 0.000000000000001, 0.000000000000001)),
}


def fundamental_for(name: str) -> np.ndarray:
    c = CALIB[name]
    return fundamental_21(c["K"], c["K_right"], c["R21"], c["T21"])


def fundamental_21(K_left, K_right, R21, T21) -> np.ndarray:
    """F21 = K_r^-T [T21]x R21 K_l^-1, as src/Dataset.cpp:106 forms it."""
    def kmat(k):
        return np.array([[k[0], 0.0, k[2]], [0.0, k[1], k[3]], [0.0, 0.0, 1.0]])
    t = np.asarray(T21, dtype=np.float64)
    tx = np.array([[0.0, -t[2], t[1]], [t[2], 0.0, -t[0]], [-t[1], t[0], 0.0]])
    return np.linalg.inv(kmat(K_right)).T @ (tx @ np.asarray(R21, dtype=np.float64)) @ np.linalg.inv(kmat(K_left))
