"""Oracle for decode + rotated NMS.  The reference's IoU/NMS is CUDA only, so there is no reference
output to pin against (parity unpinned, stated in oracle/box_nms.c); the C restatement is checked here against an
independent float64 polygon clipping of the same rectangles, and the greedy selection against its definition."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lb = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    lb.ov_iou_bev.restype = C.c_float
    return lb


def rand_boxes(n, seed, spread=6.0):
    r = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = r.uniform(-spread, spread, (n, 2))
    b[:, 2] = r.uniform(-1, 1, n)
    b[:, 3:6] = r.uniform(0.6, 5.0, (n, 3))
    b[:, 6] = r.uniform(-np.pi, np.pi, n)
    return b


def rect(b):
    c, s = np.cos(np.float64(b[6])), np.sin(np.float64(b[6]))
    loc = np.array([[-b[3], -b[4]], [b[3], -b[4]], [b[3], b[4]], [-b[3], b[4]]], np.float64) / 2
    return loc @ np.array([[c, s], [-s, c]]) + np.array([b[0], b[1]], np.float64)


def clip_area(pa, pb):
    """Sutherland-Hodgman clipping of convex polygon pa by convex polygon pb (both counter-clockwise), float64"""
    out = [tuple(p) for p in pa]
    for k in range(len(pb)):
        a, b = pb[k], pb[(k + 1) % len(pb)]
        inp, out = out, []
        if not inp:
            break
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])  # noqa: E731
        for i in range(len(inp)):
            p, q = inp[i], inp[(i + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if sp * sq < 0:
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    if len(out) < 3:
        return 0.0
    x, y = np.array([p[0] for p in out]), np.array([p[1] for p in out])
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def test_c_iou_matches_independent_clipping(lib):
    a, b = rand_boxes(60, 1), rand_boxes(60, 2)
    out = np.empty((60, 60), np.float32)
    lib.ov_iou_bev_matrix(a.ctypes.data_as(C.POINTER(C.c_float)), 60, b.ctypes.data_as(C.POINTER(C.c_float)), 60,
                          out.ctypes.data_as(C.POINTER(C.c_float)))
    ref = np.zeros((60, 60))
    for i in range(60):
        for j in range(60):
            inter = clip_area(rect(a[i]), rect(b[j]))
            ref[i, j] = inter / max(a[i, 3] * a[i, 4] + b[j, 3] * b[j, 4] - inter, 1e-8)
    assert (ref > 0.05).sum() > 50                       # the sample really contains overlapping pairs
    # the reference algorithm counts corners within a 1e-2 margin as inside: tiny positive bias on grazing contacts
    assert np.abs(out - ref).max() < 2e-2 and np.abs(out - ref).mean() < 2e-4
    same = np.empty((60, 60), np.float32)
    lib.ov_iou_bev_matrix(a.ctypes.data_as(C.POINTER(C.c_float)), 60, a.ctypes.data_as(C.POINTER(C.c_float)), 60,
                          same.ctypes.data_as(C.POINTER(C.c_float)))
    np.testing.assert_allclose(np.diag(same), 1.0, atol=1e-4)


def test_c_greedy_nms_definition(lib):
    b = rand_boxes(300, 3, spread=10.0)
    keep = np.empty(300, np.int64)
    n = lib.ov_nms_sorted(b.ctypes.data_as(C.POINTER(C.c_float)), 300, C.c_float(0.2), keep.ctypes.data_as(C.POINTER(C.c_int64)))
    keep = keep[:n]
    assert 10 < n < 300 and keep[0] == 0 and (np.diff(keep) > 0).all()
    iou = np.empty((300, 300), np.float32)
    lib.ov_iou_bev_matrix(b.ctypes.data_as(C.POINTER(C.c_float)), 300, b.ctypes.data_as(C.POINTER(C.c_float)), 300,
                          iou.ctypes.data_as(C.POINTER(C.c_float)))
    kept = set(keep.tolist())
    for i in range(300):
        earlier = [k for k in keep if k < i]
        suppressed = any(iou[k, i] > 0.2 for k in earlier)
        assert (i in kept) == (not suppressed)


def test_decode_polar_geometry():
    """a single confident cell decodes to its Cartesian centre + offset; exp / atan2 / sigmoid as in center_head.py:350-402"""
    from oracle import polar_oracle as O
    H = W = 8
    z = lambda c: np.zeros((1, H, W, c), np.float32)  # noqa: E731
    p = dict(hm=z(3) - 9.0, reg=z(2), height=z(1), dim=z(3), rot=z(2), vel=z(2))
    p["hm"][0, 2, 5, 1] = 4.0
    p["reg"][0, 2, 5] = (0.25, -0.5)
    p["rot"][0, 2, 5] = (np.sin(0.7), np.cos(0.7))
    p["dim"][0, 2, 5] = np.log([4.0, 2.0, 1.5])
    boxes, hm = O.center_decode(p, "cylinder", 4, [0.1, 0.01, 8], [0.3, -3.14, -5, 50, 3.14, 3])
    cell = 2 * W + 5
    rho, az = 5 * 4 * 0.1 + 0.3, 2 * 4 * 0.01 - 3.14
    np.testing.assert_allclose(boxes[0, cell, :2], [rho * np.cos(az) + 0.25, rho * np.sin(az) - 0.5], rtol=1e-5)
    np.testing.assert_allclose(boxes[0, cell, 3:6], [4.0, 2.0, 1.5], rtol=1e-5)
    assert abs(boxes[0, cell, -1] - 0.7) < 1e-5 and hm[0, cell].argmax() == 1 and abs(hm[0, cell, 1] - 1 / (1 + np.exp(-4.0))) < 1e-6
