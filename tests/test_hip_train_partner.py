"""One training iteration of the PARTNER detector (VoxelNetV3: mean VFE -> SpMiddleResNetFHD -> 2 x SetBlock -> RPN ->
E2ESWVoteHead + set criterion) on a reduced model at batch 2: loss terms and the gradient of EVERY parameter against fp64 autograd
over the composition of the oracle restatements (oracle/polar_oracle.py, oracle/e2e_loss_oracle.py), then the optimizer step.
Parity of the sparse convolutions, of the head and of the IoU term is unpinned by the reference (spconv absent, SURVEY F3,
CUDA-only IoU); the rest of the chain is pinned through the oracle's goldens.
Tolerances: loss terms 3e-4 relative; gradients, as max|got - ref| / max|ref| per parameter tensor (fp32 kernels through ~70 layers
and a BatchNorm over 256 cells against fp64; the fp32 rotated-IoU target of the IoU branch, itself 2e-4 from the oracle's,
adds a common-mode ~2e-3 to every upstream gradient): median < 4e-3, fewer than 10 % of the tensors above 5e-3, none above 3e-2; tensors whose
true gradient vanishes are measured against 1e-2 of the median gradient magnitude."""
import logging

import numpy as np
import pytest
import torch

from partner_amd.utils import synth
from tests.test_hip_sparse import random_voxels

pytestmark = pytest.mark.gpu

SHAPE = [64, 128, 40]          # x (range), y (azimuth), z -> BEV 8 x 16 after the /8 encoder, D' = 2 -> 256 channels
NECK = dict(layer_nums=[1, 1], ds_layer_strides=[1, 2], ds_num_filters=[32, 64], us_layer_strides=[1, 2], us_num_filters=[32, 32],
            num_input_features=256)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from partner_amd import hip
    hip.load()
    return torch.device("cuda:0")


def build(dev):
    import partner_amd as P
    from partner_amd.attention import SetBlock, waymo_bev_pos
    from tools.extra_configs_cfg import waymo_head_cfg
    hc = waymo_head_cfg()
    hc["in_channels"] = 64
    hc["GT_PROCESSOR_CONFIG"]["grid_size"] = SHAPE
    m = P.build_detector(dict(type="VoxelNetV3", pretrained=None, reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5),
                              backbone=dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8),
                              neck=dict(type="RPN", logger=logging.getLogger("RPN"), **NECK), bbox_head=hc, seg_head=None),
                         train_cfg=None, test_cfg=None)
    pos = waymo_bev_pos(x_size=8, y_size=16, voxel_size=(74.88 / 64, 6.28736 / 128, 0.15), scale=8)
    m.bev_pos = pos
    m.attns = torch.nn.ModuleList([SetBlock(in_dim=256, embed_dim_scale=1, num_heads=4, reso=(8, 16), mlp_ratio=4.0, qkv_bias=True, H_sp=8, W_sp=1,
                                            H=4, W=8, pos=pos, shift=(i % 2 == 1)) for i in range(2)])
    grid = m.bbox_head.offset_grid.clone()
    synth.load_filled(m, base_seed=5)
    with torch.no_grad():
        m.bbox_head.offset_grid.copy_(grid)
    return m, pos, grid


def make_example(dev, seed=3):
    feats, coors = random_voxels(2, SHAPE, 2500, 5, seed=seed)
    r = np.random.default_rng(seed)
    num = r.integers(1, 6, len(coors)).astype(np.int32)
    voxels = np.zeros((len(coors), 5, 5), np.float32)
    for i, n in enumerate(num):
        voxels[i, :n] = feats[i] + r.standard_normal((n, 5)).astype(np.float32) * 0.1
    gbox = synth.synth_vehicle_boxes(2, 6, seed=seed + 1)
    ex = dict(voxels=torch.from_numpy(voxels).to(dev), coordinates=torch.from_numpy(coors).to(dev), num_points=torch.from_numpy(num).to(dev),
              num_voxels=[int((coors[:, 0] == b).sum()) for b in range(2)], shape=[np.array(SHAPE)] * 2, global_box=torch.from_numpy(gbox))
    return ex, voxels, num, coors, gbox


def test_partner_training_step_matches_fp64_autograd(dev):
    from oracle import e2e_loss_oracle as E
    from oracle import polar_oracle as O
    from partner_amd.train_partner import PartnerTrainStep
    from tools.extra_configs_cfg import waymo_head_cfg
    m, pos, grid = build(dev)
    sd64 = {k: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k
                                                          and not k.endswith("offset_grid") and not k.endswith("xy_offset"))
            for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    ex, voxels, num, coors, gbox = make_example(dev)
    step = PartnerTrainStep(m, total_steps=100, drop=0.0, attn_drop=0.0, drop_path=0.0)
    p_before = step.ps.flat_p.clone()
    losses = step.forward_backward(ex)
    tops = [blk.last_top_idx.long().cpu() for blk in m.attns]
    inds = m.bbox_head.last_loss["indices"]

    # ---- the same iteration in fp64 over the oracle
    mean = torch.from_numpy(voxels.astype(np.float64).sum(1) / num[:, None].astype(np.float64))
    bev = O.sp_middle_resnet_fhd(sd64, "backbone.", mean, coors, 2, SHAPE, train=True)       # (B, 256, theta 16, r 8)
    B, C, TH, R = bev.shape
    tok = bev.permute(0, 1, 3, 2).reshape(B, C, R * TH).permute(0, 2, 1)                       # range-major tokens
    for i in range(2):
        sdi = {k[len(f"attns.{i}."):]: v for k, v in sd64.items() if k.startswith(f"attns.{i}.")}
        tok = O.set_attention(sdi, "attns.", tok, pos[..., :2].double().repeat(B, 1, 1, 1), (R, TH), 4, 4, 8, i % 2 == 1, top_override=tops[i],
                              train=True)
    x = tok.permute(0, 2, 1).reshape(B, C, R, TH).permute(0, 1, 3, 2)
    x2 = O.rpn(sd64, "neck.", x, training=True, **NECK)
    preds = O.e2e_swv_head(sd64, "bbox_head.", x2, grid.double(), window=7, depth=2, heads=4, iou=True, train=True)
    gt_cfg = waymo_head_cfg()["GT_PROCESSOR_CONFIG"]
    kw = dict(max_space=gt_cfg["max_volumn_space"], min_space=gt_cfg["min_volumn_space"], grid_size=SHAPE, stride=8,
              gaussian_overlap=gt_cfg["gaussian_overlap"], num_max_objs=gt_cfg["num_max_objs"])
    ref, _ = E.e2e_swv_loss({k: preds[k] for k in ("hm", "reg", "height", "dim", "rot", "iou", "pred_centers", "pred_vote_cls")},
                            torch.from_numpy(gbox), grid.double(), ["Vehicle"], {"Vehicle": 1}, kw, iou=True, indices=inds)
    pairs = dict(det_loss="loss", ce_loss="loss_ce", bbox_loss="loss_bbox", vote_reg_loss="loss_vote", vote_cls_loss="loss_vote_cls", iou_loss="loss_iou")
    for k, rk in pairs.items():
        np.testing.assert_allclose(float(losses[k][0]), float(ref[rk].detach()), rtol=3e-4, atol=1e-6, err_msg=k)
    ref["loss"].backward()
    grads = {n: step.ps.g[n] for n in step.ps.names}
    scale = float(np.median([float(p.grad.abs().max()) for p in sd64.values() if p.requires_grad and p.grad is not None]))
    worst, unused = {}, []
    for name, p64 in sd64.items():
        if not p64.requires_grad:
            continue
        if p64.grad is None:
            unused.append(name)
            assert float(grads[name].abs().max()) == 0.0, name
            continue
        worst[name] = float((grads[name].double().cpu() - p64.grad).abs().max() / max(float(p64.grad.abs().max()), 1e-2 * scale))
    assert all(".attns.pos_embedding_cart." in n for n in unused), unused   # constructed and never used by the reference either
    assert len(worst) > 250
    vals = np.array(list(worst.values()))
    bad = {k: v for k, v in worst.items() if v > 3e-2}
    assert not bad, (len(bad), sorted(bad.items(), key=lambda kv: -kv[1])[:8])
    assert float(np.median(vals)) < 4e-3 and float((vals > 5e-3).mean()) < 0.10, (float(np.median(vals)), float((vals > 5e-3).mean()))
    # ---- optimizer step: parameters move, loss of a second iteration on the same batch goes down
    step.optimizer_step()
    assert not torch.equal(step.ps.flat_p, p_before)
    l0 = float(losses["det_loss"][0])
    for _ in range(3):
        l1 = float(step.step(ex)["det_loss"][0])
    assert np.isfinite(l1) and l1 < l0, (l0, l1)


def test_partner_training_step_full_size_waymo_config(dev):
    """the config that IS PARTNER (configs/waymo/polar_partner_c4.py = the model section of the reference's
    waymo_partner_36epoch.py) takes training iterations at its per-GPU batch of 2: two synthetic 180k-point sweeps, 144 x 256 x 256
    token maps through both SetBlocks, the 256 x 144 head map.  Checks: finite loss terms, a gradient for every used parameter,
    determinism of the iteration (dropout off), and that repeated steps on the same batch lower the loss."""
    import os
    import time
    import partner_amd as P
    from partner_amd.train_partner import PartnerTrainStep
    from partner_amd.voxel_generator import VoxelGenerator
    cfg_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "waymo", "polar_partner_c4.py")
    w = P.Config.fromfile(cfg_path)
    m = P.build_detector(w.model, train_cfg=w.train_cfg, test_cfg=None)
    geo = {k: getattr(m.bbox_head, k).clone() for k in ("offset_grid", "xy_offset")}
    synth.load_filled(m, base_seed=31)
    for k, v in geo.items():
        getattr(m.bbox_head, k).data.copy_(v)
    m = m.to(dev).train()
    vg = VoxelGenerator(synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
    vs, cs, ns, counts = [], [], [], []
    for b in range(2):
        sw = torch.from_numpy(synth.synth_sweep_beams_polar(180000, seed=b)).to(dev)
        voxels, coors, num = vg.generate(sw)
        vs.append(voxels)
        ns.append(num)
        cs.append(torch.cat([torch.full((coors.shape[0], 1), b, dtype=coors.dtype, device=dev), coors], 1))
        counts.append(int(voxels.shape[0]))
    gbox = synth.synth_vehicle_boxes(2, 40, seed=2)
    ex = dict(voxels=torch.cat(vs), coordinates=torch.cat(cs), num_points=torch.cat(ns), num_voxels=counts, shape=[np.array([1152, 2048, 40])] * 2,
              global_box=torch.from_numpy(gbox))
    step = PartnerTrainStep(m, total_steps=100, drop=0.0, attn_drop=0.0, drop_path=0.0)
    l0 = step.forward_backward(ex)
    g0 = step.ps.flat_g.clone()
    for k, v in l0.items():
        assert np.isfinite(float(v[0])), k
    used = [n for n in step.ps.names if ".attns.pos_embedding_cart." not in n]
    zero = [n for n in used if float(step.ps.g[n].abs().max()) == 0.0]
    # a conv bias directly in front of BatchNorm has a vanishing (not structurally zero) gradient; nothing else may be all-zero
    assert not [n for n in zero if not n.endswith(".bias")], zero[:8]
    assert torch.isfinite(step.ps.flat_g).all()
    l0b = step.forward_backward(ex)
    assert float(l0b["det_loss"][0]) == float(l0["det_loss"][0])
    # BatchNorm running statistics moved between the two passes, the parameters did not: gradients agree to rounding of the sums only
    assert float((step.ps.flat_g - g0).abs().max()) <= 1e-5 * float(g0.abs().max())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        l1 = step.step(ex)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"PARTNER bs=2 training iteration: {ms:.1f} ms")
    assert float(l1["det_loss"][0]) < float(l0["det_loss"][0])


def _ddp_rank(rank, world, port, q):
    """one rank of a 2-rank PartnerTrainStep.step (both ranks on cuda:0, gloo transport): rank-specific start, rank-specific clouds"""
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from partner_amd import dist_utils as D
    from partner_amd.train_partner import PartnerTrainStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    assert D.init("gloo") is True
    m, _, _ = build(dev)
    m = m.to(dev).train()
    step = PartnerTrainStep(m, total_steps=100, drop=0.0, attn_drop=0.0, drop_path=0.0)
    if rank == 1:
        with torch.no_grad():
            step.ps.flat_p.mul_(1.5)
    step.sync_initial_params()
    start = step.ps.flat_p.clone()
    ex = make_example(dev, seed=3 + 10 * rank)[0]
    ex["global_box"] = make_example(dev, seed=3)[0]["global_box"]     # the same boxes: the ranks' matched counts (num_boxes) agree
    losses = step.step(ex)
    q.put((rank, start.cpu().numpy(), step.ps.flat_p.cpu().numpy(), float(losses["det_loss"][0])))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_partner_two_rank_step_matches_single_process_mean(dev):
    """world size 2: equal parameters on both ranks after sync_initial_params and after the step, equal to ONE process applying
    the mean of the two ranks' gradients with the same optimizer step (the reference averages gradients over ranks, dist_utils.py:17-28)"""
    import socket
    import torch.multiprocessing as mp
    from partner_amd.train_partner import PartnerTrainStep
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    assert res[0][3] != res[1][3]                                       # the ranks really saw different clouds
    m, _, _ = build(dev)
    m = m.to(dev).train()
    step = PartnerTrainStep(m, total_steps=100, drop=0.0, attn_drop=0.0, drop_path=0.0)
    np.testing.assert_array_equal(step.ps.flat_p.cpu().numpy(), res[0][1])
    gb = make_example(dev, seed=3)[0]["global_box"]
    total = None
    for rank in range(2):
        ex = make_example(dev, seed=3 + 10 * rank)[0]
        ex["global_box"] = gb
        step.forward_backward(ex, grad_scale=0.5)
        total = step.ps.flat_g.clone() if total is None else total + step.ps.flat_g
    step.ps.flat_g.copy_(total)
    step.optimizer_step()
    np.testing.assert_allclose(step.ps.flat_p.cpu().numpy(), res[0][2], rtol=2e-5, atol=2e-7)
