"""GPU parity of decode + rotated NMS against the oracle (numpy decode + the C restatement of the
BEV rotated IoU / greedy NMS).  The reference's own IoU kernel is CUDA only: parity unpinned by the reference."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from partner_amd.utils import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "no GPU visible"
    from partner_amd import hip
    hip.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def clib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))


def synth_head_outputs(b, h, w, ncls, n_obj, seed, with_vel=True):
    """raw head tensors (NHWC) with clusters of confident, overlapping detections around n_obj objects per sample"""
    r = np.random.default_rng(seed)
    p = dict(hm=(r.standard_normal((b, h, w, ncls)) * 0.3 - 6.0).astype(np.float32), reg=r.uniform(-0.3, 0.3, (b, h, w, 2)).astype(np.float32),
             height=r.uniform(-2, 1, (b, h, w, 1)).astype(np.float32), dim=r.uniform(-0.3, 0.3, (b, h, w, 3)).astype(np.float32),
             rot=r.standard_normal((b, h, w, 2)).astype(np.float32))
    if with_vel:
        p["vel"] = r.standard_normal((b, h, w, 2)).astype(np.float32)
    for i in range(b):
        for _ in range(n_obj):
            cy, cx, c = r.integers(2, h - 2), r.integers(2, w - 2), r.integers(0, ncls)
            ang, size = r.uniform(-np.pi, np.pi), np.log(r.uniform(1.5, 5.0, 3))
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    p["hm"][i, cy + dy, cx + dx, c] = r.uniform(-1.0, 3.0)
                    p["dim"][i, cy + dy, cx + dx] = size + r.normal(0, 0.05, 3)
                    p["rot"][i, cy + dy, cx + dx] = (np.sin(ang) + r.normal(0, 0.05), np.cos(ang) + r.normal(0, 0.05))
    return p


@pytest.mark.parametrize("cfg", [dict(shape="cylinder", rectify=False, vel=True), dict(shape="cylinder", rectify=True, vel=True),
                                 dict(shape="cuboid", rectify=False, vel=False)], ids=lambda c: f"{c['shape']}-rect{int(c['rectify'])}-vel{int(c['vel'])}")
def test_predict_matches_oracle(dev, clib, cfg):
    import partner_amd as P
    from oracle import polar_oracle as O
    from partner_amd import ops
    from tests.test_oracle_golden import TASKS
    b, h, w, ncls = 2, 64, 64, 10
    p = synth_head_outputs(b, h, w, ncls, 40, seed=7 + int(cfg["rectify"]), with_vel=cfg["vel"])
    vs, pr, osf = [0.4, 0.05, 8.0], [0.3, -1.6, -5.0, 50.0, 1.6, 3.0], 2
    if cfg["shape"] == "cuboid":
        vs, pr = [0.4, 0.4, 8.0], [-25.6, -25.6, -5.0, 25.6, 25.6, 3.0]
    test_cfg = dict(post_center_limit_range=[-60.0, -60.0, -10.0, 60.0, 60.0, 10.0], score_threshold=0.1, out_size_factor=osf, voxel_size=vs,
                    pc_range=pr, rectify=cfg["rectify"], nms=dict(nms_pre_max_size=300, nms_post_max_size=83, nms_iou_threshold=0.2))
    head = P.build_bbox_head(dict(type="CenterHead", in_channels=32, tasks=TASKS, dataset="nuscenes", weight=0.25, code_weights=[1.0] * 10,
                                  common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                                  voxel_shape=cfg["shape"]))
    preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
    example = dict(metadata=["a", "b"], pc_range=np.stack([np.float32(pr)] * b))
    got = head.predict(example, preds, test_cfg)
    assert len(got) == b and got[1]["metadata"] == "b"
    # oracle
    boxes, hm = O.center_decode(p, cfg["shape"], osf, vs, pr, rectify=cfg["rectify"])

    def c_nms(sorted_boxes, thr):
        keep = np.empty(len(sorted_boxes), np.int64)
        sb = np.ascontiguousarray(sorted_boxes, np.float32)
        n = clib.ov_nms_sorted(sb.ctypes.data_as(C.POINTER(C.c_float)), len(sb), C.c_float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
        return keep[:n]

    for i in range(b):
        ref = O.center_post_process(boxes[i], hm[i], 0.1, test_cfg["post_center_limit_range"], 0.2, 300, 83, c_nms)
        g = got[i]
        assert 20 < len(ref["cells"]) <= 83
        np.testing.assert_array_equal(g["cells"].cpu().numpy(), ref["cells"])
        np.testing.assert_array_equal(g["label_preds"].cpu().numpy(), ref["label_preds"])
        np.testing.assert_allclose(g["scores"].cpu().numpy(), ref["scores"], rtol=1e-5, atol=1e-7)
        gb, rb = g["box3d_lidar"].cpu().numpy(), ref["box3d_lidar"]
        assert gb.shape == rb.shape and gb.shape[1] == (9 if cfg["vel"] else 7)
        d = np.abs(gb - rb)
        d[:, -1] = np.minimum(d[:, -1], np.abs(d[:, -1] - 2 * np.pi))   # the heading may differ by a full turn at +-pi
        assert d.max() < 2e-4


def test_predict_per_class_nms(dev, clib):
    """test_cfg.per_class_nms = True (the reference's nuScenes configs): objects of DIFFERENT classes on top of each other all
    survive, duplicates of the same class are suppressed; cells / labels / order against the oracle"""
    import partner_amd as P
    from oracle import polar_oracle as O
    from tests.test_oracle_golden import TASKS
    b, h, w, ncls = 2, 64, 64, 10
    p = synth_head_outputs(b, h, w, ncls, 60, seed=41, with_vel=True)
    # stack a second and a third class onto the clusters: same cells, other channels
    r = np.random.default_rng(3)
    strong = p["hm"].max(-1) > -1.0
    for c_shift in (3, 7):
        ys, xs = np.nonzero(strong[0])
        for y, x in zip(ys[::2], xs[::2]):
            x2 = min(w - 1, x + 1)                           # the neighbouring cell: an overlapping box of another class
            p["hm"][0, y, x2, :] = -6.0
            p["hm"][0, y, x2, (int(p["hm"][0, y, x].argmax()) + c_shift) % ncls] = r.uniform(0.0, 3.0)
            p["dim"][0, y, x2] = p["dim"][0, y, x]
            p["rot"][0, y, x2] = p["rot"][0, y, x]
    vs, pr, osf = [0.4, 0.05, 8.0], [0.3, -1.6, -5.0, 50.0, 1.6, 3.0], 2
    test_cfg = dict(post_center_limit_range=[-60.0, -60.0, -10.0, 60.0, 60.0, 10.0], score_threshold=0.1, out_size_factor=osf, voxel_size=vs,
                    pc_range=pr, rectify=False, per_class_nms=True, max_per_img=500,
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.1))
    head = P.build_bbox_head(dict(type="CenterHead", in_channels=32, tasks=TASKS, dataset="nuscenes", weight=0.25, code_weights=[1.0] * 10,
                                  common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                                  voxel_shape="cylinder"))
    preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
    got = head.predict(dict(metadata=["a", "b"]), preds, test_cfg)
    plain = head.predict(dict(metadata=["a", "b"]), preds, dict(test_cfg, per_class_nms=False))
    boxes, hm = O.center_decode(p, "cylinder", osf, vs, pr, rectify=False)

    def c_nms(sorted_boxes, thr):
        keep = np.empty(len(sorted_boxes), np.int64)
        sb = np.ascontiguousarray(sorted_boxes, np.float32)
        n = clib.ov_nms_sorted(sb.ctypes.data_as(C.POINTER(C.c_float)), len(sb), C.c_float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
        return keep[:n]

    for i in range(b):
        ref = O.center_post_process(boxes[i], hm[i], 0.1, test_cfg["post_center_limit_range"], 0.1, 4096, 83, c_nms, per_class=True)
        np.testing.assert_array_equal(got[i]["cells"].cpu().numpy(), ref["cells"])
        np.testing.assert_array_equal(got[i]["label_preds"].cpu().numpy(), ref["label_preds"])
        np.testing.assert_allclose(got[i]["scores"].cpu().numpy(), ref["scores"], rtol=1e-5, atol=1e-7)
    # the class-aware pass keeps overlapping boxes of different classes that the multi-class pass removes
    assert len(got[0]["cells"]) > len(plain[0]["cells"]) or len(got[0]["cells"]) == 83


@pytest.mark.parametrize("rectify", [False, True])
def test_swv_head_predict_matches_oracle(dev, clib, rectify):
    """E2ESWVoteHead.predict (the Waymo PARTNER config's head: IoU-rectified scores, Cartesian reg + offset grid, one class) with the
    config's own test_cfg sizes (pre 4096 / post 500 / IoU 0.7) against the oracle restatement"""
    import os
    import partner_amd as P
    from oracle import polar_oracle as O
    cfg = P.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "waymo", "polar_partner_c4.py"))
    head = P.build_bbox_head(cfg.model["bbox_head"] if isinstance(cfg.model, dict) else cfg.model.bbox_head).to(dev).eval()
    h, w = 256, 144
    p = synth_head_outputs(2, h, w, 1, 120, seed=61, with_vel=False)
    r = np.random.default_rng(9)
    p["iou"] = r.uniform(-1.2, 1.2, (2, h, w, 1)).astype(np.float32)
    p["reg"] = r.uniform(-0.5, 0.5, (2, h, w, 2)).astype(np.float32)
    test_cfg = dict(post_center_limit_range=[-80.0, -80.0, -10.0, 80.0, 80.0, 10.0], score_threshold=0.1, rectify=rectify,
                    nms=dict(nms_pre_max_size=4096, nms_post_max_size=500, nms_iou_threshold=0.7))
    preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
    got = head.predict(dict(metadata=["a", "b"]), preds, test_cfg)
    boxes, hm = O.swv_decode(p, head.offset_grid[0].cpu().numpy(), int(head.iou_factor), rectify)

    def c_nms(sorted_boxes, thr):
        keep = np.empty(len(sorted_boxes), np.int64)
        sb = np.ascontiguousarray(sorted_boxes, np.float32)
        n = clib.ov_nms_sorted(sb.ctypes.data_as(C.POINTER(C.c_float)), len(sb), C.c_float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
        return keep[:n]

    for i in range(2):
        ref = O.center_post_process(boxes[i], hm[i], 0.1, test_cfg["post_center_limit_range"], 0.7, 4096, 500, c_nms)
        assert len(ref["cells"]) > 50
        np.testing.assert_array_equal(got[i]["cells"].cpu().numpy(), ref["cells"])
        np.testing.assert_allclose(got[i]["scores"].cpu().numpy(), ref["scores"], rtol=1e-5, atol=1e-7)
        d = np.abs(got[i]["box3d_lidar"].cpu().numpy() - ref["box3d_lidar"])
        d[:, -1] = np.minimum(d[:, -1], np.abs(d[:, -1] - 2 * np.pi))
        assert d.max() < 2e-4
    assert got[1]["metadata"] == "b"


def test_detector_predict_with_config_test_cfg(dev):
    """the nuScenes config's own test_cfg (per_class_nms, rectify) through the detector: forward -> bbox_head.predict"""
    import os
    import partner_amd as P
    from partner_amd import ops
    cfg = P.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "nusc", "polar_pillar_partner_c2.py"))
    assert cfg.test_cfg["per_class_nms"] and cfg.test_cfg["nms"]["nms_iou_threshold"] == 0.1
    m = P.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.load_filled(m, base_seed=0)
    m = m.to(dev).eval()
    pts = ops.cart_to_polar(torch.from_numpy(synth.synth_sweep_cart(30000, seed=2)).to(dev))
    preds = m.forward_points(pts, torch.tensor([0, 30000], dtype=torch.int32, device=dev), 1)
    preds["hm"] = preds["hm"] + 2.0                       # random-init weights give no peaks: lift the logits over the threshold
    out = m.bbox_head.predict(dict(metadata=["tok"]), {"det_preds": [preds]}, cfg.test_cfg)
    assert len(out) == 1 and out[0]["metadata"] == "tok"
    n = out[0]["scores"].numel()
    assert 0 < n <= 83 and out[0]["box3d_lidar"].shape == (n, 9) and torch.isfinite(out[0]["box3d_lidar"]).all()
    s = out[0]["scores"]
    assert (s[:-1] >= s[1:]).all()
    # the detector's own forward(example, return_loss=False) ends in predict when it was built with a test_cfg (point_pillars.py:104-108)
    spec = ops.GridSpec.from_range(synth.NUSC_RANGE, synth.NUSC_VOXEL)
    gi, _ = ops.grid_index(pts, torch.tensor([0, 30000], dtype=torch.int32, device=dev), 1, spec)
    example = dict(points=pts, grid_ind=gi, num_points=[30000], voxel_size=np.stack([np.float32(synth.NUSC_VOXEL)]),
                   pc_range=np.stack([np.float32(synth.NUSC_RANGE)]), grid_size=np.stack([np.array([512, 512, 1])]), metadata=["tok"])
    det = m(example, return_loss=False)
    assert set(det) == {"det"} and len(det["det"]) == 1 and set(det["det"][0]) >= {"box3d_lidar", "scores", "label_preds", "metadata"}
    raw = m(example, return_loss=False, raw_preds=True)
    assert "det_preds" in raw


def test_predict_large_pre_max(dev, clib):
    """Waymo-style post-processing sizes: thousands of candidates, pre_max 1024 (histogram cut far from the top bins), post 300,
    IoU 0.7 -- same cells, in the same order, as the oracle"""
    import partner_amd as P
    from oracle import polar_oracle as O
    from tests.test_oracle_golden import TASKS
    b, h, w, ncls = 1, 96, 128, 3
    p = synth_head_outputs(b, h, w, ncls, 400, seed=23, with_vel=False)
    p["hm"] += np.random.default_rng(5).uniform(0.0, 4.5, p["hm"].shape).astype(np.float32)   # a broad score distribution
    vs, pr, osf = [0.2, 0.02, 8.0], [0.3, -1.6, -5.0, 50.0, 1.6, 3.0], 2
    test_cfg = dict(post_center_limit_range=[-80.0, -80.0, -10.0, 80.0, 80.0, 10.0], score_threshold=0.1, out_size_factor=osf, voxel_size=vs,
                    pc_range=pr, rectify=True, nms=dict(nms_pre_max_size=1024, nms_post_max_size=300, nms_iou_threshold=0.7))
    head = P.build_bbox_head(dict(type="CenterHead", in_channels=32, tasks=[dict(num_class=3, class_names=["a", "b", "c"])], dataset="waymo", weight=0.25,
                                  code_weights=[1.0] * 8, common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)},
                                  voxel_shape="cylinder"))
    preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
    got = head.predict(dict(metadata=[None]), preds, test_cfg)[0]
    boxes, hm = O.center_decode(p, "cylinder", osf, vs, pr, rectify=True)

    def c_nms(sorted_boxes, thr):
        keep = np.empty(len(sorted_boxes), np.int64)
        sb = np.ascontiguousarray(sorted_boxes, np.float32)
        n = clib.ov_nms_sorted(sb.ctypes.data_as(C.POINTER(C.c_float)), len(sb), C.c_float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
        return keep[:n]

    ref = O.center_post_process(boxes[0], hm[0], 0.1, test_cfg["post_center_limit_range"], 0.7, 1024, 300, c_nms)
    assert int((hm[0].max(-1) > 0.1).sum()) > 3000 and len(ref["cells"]) == 300
    np.testing.assert_array_equal(got["cells"].cpu().numpy(), ref["cells"])
    np.testing.assert_array_equal(got["label_preds"].cpu().numpy(), ref["label_preds"])
    np.testing.assert_allclose(got["scores"].cpu().numpy(), ref["scores"], rtol=1e-5, atol=1e-7)


def test_predict_edge_cases(dev):
    """no detection above the threshold; every cell above it (candidate cap / pre_max truncation)"""
    import partner_amd as P
    from tests.test_oracle_golden import TASKS
    head = P.build_bbox_head(dict(type="CenterHead", in_channels=32, tasks=TASKS, dataset="nuscenes", weight=0.25, code_weights=[1.0] * 10,
                                  common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}, voxel_shape="cylinder"))
    test_cfg = dict(post_center_limit_range=[-60.0, -60.0, -10.0, 60.0, 60.0, 10.0], score_threshold=0.1, out_size_factor=4,
                    voxel_size=[0.098, 0.0123, 8.0], pc_range=[0.3, -3.1488, -5.0, 50.476, 3.1488, 3.0],
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2))
    p = synth_head_outputs(1, 128, 128, 10, 0, seed=1)
    preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
    out = head.predict(dict(metadata=[None]), preds, test_cfg)
    assert out[0]["box3d_lidar"].shape == (0, 9) and out[0]["scores"].numel() == 0
    p["hm"] += 9.0   # 16384 candidates: more than the 8192-entry sort buffer
    preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
    out = head.predict(dict(metadata=[None]), preds, test_cfg)
    n = out[0]["scores"].numel()
    assert 0 < n <= 83 and torch.isfinite(out[0]["box3d_lidar"]).all()
    s = out[0]["scores"]
    assert (s[:-1] >= s[1:]).all()


def test_predict_near_constant_map_keeps_the_late_peaks(dev, clib):
    """more candidates above the threshold than the sort buffer holds (an untrained head: heat-map bias -2.19 -> sigmoid 0.1 on all of a
    160 x 160 map = 25 600 cells > 8192) and a few clearly better cells LATE in cell order: the selection must still be the best
    `pre_max` by score (ties by cell index), as the reference's full sort gives -- not the first cells in cell order"""
    import partner_amd as P
    from oracle import polar_oracle as O
    from tests.test_oracle_golden import TASKS
    b, h, w, ncls = 1, 160, 160, 10
    r = np.random.default_rng(5)
    p = dict(hm=np.full((b, h, w, ncls), -6.0, np.float32), reg=r.uniform(-0.3, 0.3, (b, h, w, 2)).astype(np.float32),
             height=r.uniform(-2, 1, (b, h, w, 1)).astype(np.float32), dim=r.uniform(-0.3, 0.3, (b, h, w, 3)).astype(np.float32),
             rot=r.standard_normal((b, h, w, 2)).astype(np.float32), vel=r.standard_normal((b, h, w, 2)).astype(np.float32))
    # every cell just above the threshold and inside ONE bin of the top 12 score bits ([0.1172, 0.125)); the logits are distinct
    # multiples of 2e-6, i.e. the scores are ~28 ulp apart: the ranking does not depend on the last bit of anybody's sigmoid
    p["hm"][..., 3] = (-2.0 + 2e-6 * r.permutation(b * h * w).reshape(b, h, w)).astype(np.float32)
    peaks = r.choice(np.arange(h * w // 2, h * w), 60, replace=False)              # the good cells sit in the second half of the map
    for c in peaks:
        p["hm"][0, c // w, c % w, 7] = r.uniform(1.0, 3.0)
    vs, pr, osf = [0.16, 0.0196, 8.0], [0.3, -1.57, -5.0, 51.5, 1.57, 3.0], 2
    test_cfg = dict(post_center_limit_range=[-80.0, -80.0, -10.0, 80.0, 80.0, 10.0], score_threshold=0.1, out_size_factor=osf, voxel_size=vs,
                    pc_range=pr, rectify=False, nms=dict(nms_pre_max_size=1000, nms_post_max_size=200, nms_iou_threshold=0.2))
    head = P.build_bbox_head(dict(type="CenterHead", in_channels=32, tasks=TASKS, dataset="nuscenes", weight=0.25, code_weights=[1.0] * 10,
                                  common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}, voxel_shape="cylinder"))
    preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
    got = head.predict(dict(metadata=[None]), preds, test_cfg)[0]
    boxes, hm = O.center_decode(p, "cylinder", osf, vs, pr, rectify=False)

    def c_nms(sorted_boxes, thr):
        keep = np.empty(len(sorted_boxes), np.int64)
        sb = np.ascontiguousarray(sorted_boxes, np.float32)
        n = clib.ov_nms_sorted(sb.ctypes.data_as(C.POINTER(C.c_float)), len(sb), C.c_float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
        return keep[:n]

    ref = O.center_post_process(boxes[0], hm[0], 0.1, test_cfg["post_center_limit_range"], 0.2, 1000, 200, c_nms)
    kept = set(got["cells"].cpu().numpy().tolist())
    assert set(peaks.tolist()) & kept, "none of the high-scoring late cells survived"
    np.testing.assert_array_equal(got["cells"].cpu().numpy(), ref["cells"])
    np.testing.assert_allclose(got["scores"].cpu().numpy(), ref["scores"], rtol=1e-5, atol=1e-7)


def test_double_flip_merge_and_predict(dev, clib, golden):
    """test_cfg.double_flip (center_head.py:412, 289-346): (1) the merge kernel against the maps the reference's double_flip_decode
    produced (golden double_flip.npz; sums of four in copy order, sigmoid / exp in f32: 2 ulp), (2) predict() on a batch of two
    groups of four flipped copies against oracle merge -> decode(activated) -> NMS"""
    import partner_amd as P
    from oracle import polar_oracle as O
    from partner_amd import hip
    from tests.test_oracle_golden import TASKS
    g = golden("double_flip.npz")
    names = ["hm", "reg", "height", "dim", "rot", "vel"]
    t = {k: torch.from_numpy(g[f"in_{k}"]).to(dev) for k in names}
    mb, h, w = 2, 6, 5
    out = {k: torch.empty((mb, h, w, g[f"in_{k}"].shape[3]), dtype=torch.float32, device=dev) for k in names}
    hip.call("pn_double_flip_merge_f32", t["hm"].data_ptr(), 3, 3, t["reg"].data_ptr(), 2, t["height"].data_ptr(), 1, t["dim"].data_ptr(), 3,
             t["rot"].data_ptr(), 2, t["vel"].data_ptr(), 2, mb, h, w, out["hm"].data_ptr(), out["reg"].data_ptr(), out["height"].data_ptr(),
             out["dim"].data_ptr(), out["rot"].data_ptr(), out["vel"].data_ptr(), hip.stream())
    for k in names:
        np.testing.assert_allclose(out[k].cpu().numpy(), g[f"out_{k}"], rtol=3e-7, atol=3e-7, err_msg=k)
    # ---- predict
    b, h, w, ncls = 8, 64, 64, 10
    p = synth_head_outputs(b, h, w, ncls, 40, seed=23, with_vel=True)
    vs, pr, osf = [0.4, 0.05, 8.0], [0.3, -1.6, -5.0, 50.0, 1.6, 3.0], 2
    test_cfg = dict(post_center_limit_range=[-60.0, -60.0, -10.0, 60.0, 60.0, 10.0], score_threshold=0.1, out_size_factor=osf, voxel_size=vs,
                    pc_range=pr, rectify=False, double_flip=True, nms=dict(nms_pre_max_size=300, nms_post_max_size=83, nms_iou_threshold=0.2))
    head = P.build_bbox_head(dict(type="CenterHead", in_channels=32, tasks=TASKS, dataset="nuscenes", weight=0.25, code_weights=[1.0] * 10,
                                  common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                                  voxel_shape="cylinder"))
    preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
    got = head.predict(dict(metadata=[f"m{i}" for i in range(8)]), preds, test_cfg)
    assert len(got) == 2 and [d["metadata"] for d in got] == ["m0", "m4"]
    merged = O.double_flip_merge(p)
    boxes, hm = O.center_decode(merged, "cylinder", osf, vs, pr, rectify=False, activated=True)

    def c_nms(sorted_boxes, thr):
        keep = np.empty(len(sorted_boxes), np.int64)
        sb = np.ascontiguousarray(sorted_boxes, np.float32)
        n = clib.ov_nms_sorted(sb.ctypes.data_as(C.POINTER(C.c_float)), len(sb), C.c_float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
        return keep[:n]

    for i in range(2):
        ref = O.center_post_process(boxes[i], hm[i], 0.1, test_cfg["post_center_limit_range"], 0.2, 300, 83, c_nms)
        assert len(ref["scores"]) > 5
        np.testing.assert_array_equal(got[i]["cells"].cpu().numpy(), ref["cells"])
        np.testing.assert_array_equal(got[i]["label_preds"].cpu().numpy(), ref["label_preds"])
        np.testing.assert_allclose(got[i]["scores"].cpu().numpy(), ref["scores"], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(got[i]["box3d_lidar"].cpu().numpy(), ref["box3d_lidar"], rtol=1e-5, atol=2e-5)
    with pytest.raises(ValueError):
        bad = {"det_preds": [{k: v[:6] for k, v in preds["det_preds"][0].items()}]}
        head.predict(dict(metadata=[None] * 6), bad, test_cfg)


def test_stateful_nms_across_sectors(dev, clib):
    """test_cfg.stateful_nms (the reference's 4-sector streaming configs; center_head.py:486-501, 507-509): three sectors in a row,
    each sector's predict() fed with the previous one's detections, against the oracle chain sector by sector"""
    import partner_amd as P
    from oracle import polar_oracle as O
    from tests.test_oracle_golden import TASKS
    b, h, w, ncls = 2, 32, 64, 10
    vs, pr, osf = [0.4, 0.05, 8.0], [0.3, -0.8, -5.0, 50.0, 0.8, 3.0], 2
    interval = 1.6            # one sector spans 1.6 rad
    test_cfg = dict(post_center_limit_range=[-60.0, -60.0, -10.0, 60.0, 60.0, 10.0], score_threshold=0.1, out_size_factor=osf, voxel_size=vs,
                    pc_range=pr, rectify=False, stateful_nms=True, interval=interval,
                    nms=dict(nms_pre_max_size=300, nms_post_max_size=40, nms_iou_threshold=0.2))
    head = P.build_bbox_head(dict(type="CenterHead", in_channels=32, tasks=TASKS, dataset="nuscenes", weight=0.25, code_weights=[1.0] * 10,
                                  common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                                  voxel_shape="cylinder"))

    def c_nms(sorted_boxes, thr):
        keep = np.empty(len(sorted_boxes), np.int64)
        sb = np.ascontiguousarray(sorted_boxes, np.float32)
        n = clib.ov_nms_sorted(sb.ctypes.data_as(C.POINTER(C.c_float)), len(sb), C.c_float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
        return keep[:n]

    prev_dev, prev_ref = None, [None] * b
    for sec in range(3):
        p = synth_head_outputs(b, h, w, ncls, 25, seed=50 + sec, with_vel=True)
        preds = {"det_preds": [{k: torch.from_numpy(v).to(dev).permute(0, 3, 1, 2) for k, v in p.items()}]}
        got = head.predict(dict(metadata=["a", "b"]), preds, test_cfg, sec_id=sec, prev_dets=prev_dev)
        assert isinstance(got, list) and len(got) == 1 and len(got[0]) == b          # per task, per sample, unmerged
        boxes, hm = O.center_decode(p, "cylinder", osf, vs, pr, rectify=False)
        carried = 0
        for i in range(b):
            ref = O.center_post_process_stateful(boxes[i], hm[i], 0.1, test_cfg["post_center_limit_range"], 0.2, 300, 40, c_nms, prev_ref[i],
                                                 interval * sec, sec)
            g = got[0][i]
            assert len(ref["scores"]) <= 40 * (sec + 1)
            np.testing.assert_array_equal(g["cells"].cpu().numpy(), ref["cells"])
            np.testing.assert_array_equal(g["label_preds"].cpu().numpy(), ref["label_preds"])
            np.testing.assert_allclose(g["scores"].cpu().numpy(), ref["scores"], rtol=1e-5, atol=1e-7)
            d = np.abs(g["box3d_lidar"].cpu().numpy() - ref["box3d_lidar"])
            d[:, -1] = np.minimum(d[:, -1], np.abs(d[:, -1] - 2 * np.pi))
            assert d.max() < 2e-4
            carried += int((ref["cells"] >= h * w).sum())
            prev_ref[i] = ref
        if sec > 0:
            assert carried > 10          # detections of earlier sectors survive into the later lists
        prev_dev = got
