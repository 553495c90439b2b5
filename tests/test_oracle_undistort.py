"""orc_undistort (cv::undistort restated from the published OpenCV 4.x source; parity unpinned): zero distortion is the
identity, and with the EuRoC coefficients (config/euroc.yaml:13, :18) the fixed-point result stays within the
quantisation of 1/32 px maps + 15-bit weights of an independent floating-point evaluation of the same camera model."""
import numpy as np

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc


def _float_model(img, K, d):
    h, w = img.shape
    jj, ii = np.meshgrid(np.arange(w), np.arange(h))
    x, y = (jj - K[2]) / K[0], (ii - K[3]) / K[1]
    r2 = x * x + y * y
    kr = 1 + (d[1] * r2 + d[0]) * r2
    xd = x * kr + d[2] * 2 * x * y + d[3] * (r2 + 2 * x * x)
    yd = y * kr + d[2] * (r2 + 2 * y * y) + d[3] * 2 * x * y
    u, v = K[0] * xd + K[2], K[1] * yd + K[3]
    x0, y0 = np.floor(u).astype(int), np.floor(v).astype(int)
    a, b = u - x0, v - y0
    ok = (x0 >= 0) & (x0 < w - 1) & (y0 >= 0) & (y0 < h - 1)
    x0c, y0c = np.clip(x0, 0, w - 2), np.clip(y0, 0, h - 2)
    f = img.astype(float)
    val = ((1 - a) * (1 - b) * f[y0c, x0c] + a * (1 - b) * f[y0c, x0c + 1] + (1 - a) * b * f[y0c + 1, x0c]
           + a * b * f[y0c + 1, x0c + 1])
    return val, ok


def test_zero_distortion_is_identity():
    img = synth.s2_image(120, 200)
    K = synth.CALIB["euroc"]["K"]
    assert (orc.undistort(img, K, [0, 0, 0, 0]) == img).all()
    assert (orc.undistort(img, K, [0, 0, 0, 0, 0]) == img).all()


def test_euroc_coefficients_against_float_model():
    h, w = synth.SHAPES["euroc"]
    img = synth.s2_image(h, w)
    for K, d in ((synth.CALIB["euroc"]["K"], synth.CALIB["euroc"]["dist"]),
                 (synth.CALIB["euroc"]["K_right"], synth.CALIB["euroc"]["dist_right"])):
        u = orc.undistort(img, K, d)
        val, ok = _float_model(img, K, d)
        diff = np.abs(val - u)[ok]
        assert diff.max() < 4.0 and diff.mean() < 0.5      # S2 has steps of up to ~60 grey levels per pixel: 1/32 px of that
        assert (u != img).mean() > 0.5                      # the image really is warped
        # pixels that map outside read the constant border 0 (sampled positions beyond the image at the corners)
        assert (u[~ok | (val < 0)] >= 0).all()
