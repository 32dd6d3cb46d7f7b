#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations' stages (not the headline bench):
 C5: nuScenes polar-pillar model, 10-sweep accumulation (300k points), hipGraph per frame, latency p50 / p99
 C4 stages: Waymo PARTNER grid -- hard voxelization (180k pts, P=5, Vmax=150k) + mean VFE, 2 x SetBlock on the
            (144 x 256) x 256 BEV tokens, RPN of the Waymo config (fp32)."""
import logging, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, partner_amd as P
from partner_amd import ops
from partner_amd.engine import FrameEngine
from partner_amd.attention import SetBlock, waymo_bev_pos
from partner_amd.voxel_generator import VoxelGenerator
from partner_amd.utils import synth

dev = torch.device("cuda:0")


def timeit(fn, n=20, warm=25):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# ---- C5
m = P.build_detector(bench.c2_model_cfg()); synth.load_filled(m, 0); m = m.to(dev).eval()
eng = FrameEngine(m, 1, 300000).capture()
frames = [torch.from_numpy(synth.synth_sweep_cart(300000, seed=s, n_sweeps=10)).to(dev) for s in range(4)]
lat = []
for i in range(220):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(frames[i % 4]); torch.cuda.synchronize()
    if i >= 20:
        lat.append(1e3 * (time.perf_counter() - t0))
lat = np.sort(np.array(lat))
print(f"C5  300k-pt frame, hipGraph, 1 stream: latency p50 {lat[len(lat)//2]:.3f} ms  p99 {lat[int(len(lat)*0.99)]:.3f} ms  ({1e3/lat[len(lat)//2]:.0f} frames/s)")

# ---- scatter stage alone (V0..V5: cart->polar, grid index, unique-rank, bucket, PFN, canvas) on the config grid and on the
#      BASELINE.json synthetic grid (0.3125 m x 0.05 rad: R=160, Theta=126); algorithmic bytes per SURVEY 8(d):
#      N*F*4 + V*4*8 + V*C*4 + B*C*Theta*R*4 (grid indices computed in-kernel)
def scatter_row(tag, rng_, vs):
    cfg = bench.c2_model_cfg()
    cfg["reader"].update(voxel_size=list(vs), pc_range=list(rng_))
    # (the head is not run here and keeps the config grid's position encoding)
    ms = P.build_detector(cfg); synth.load_filled(ms, 0); ms = ms.to(dev).eval()
    spec = ops.GridSpec.from_range(rng_, vs)
    cart = torch.from_numpy(synth.synth_sweep_cart(30000, seed=5)).to(dev)
    offs = torch.tensor([0, 30000], dtype=torch.int32, device=dev)
    persistent = ms.new_canvas(1, spec)
    def run():   # what a frame of the engine does before / after the backbone: encode into the persistent canvas, sparse clear
        polar = ops.cart_to_polar(cart)
        _, keys = ops.grid_index(polar, offs, 1, spec, want_grid_ind=False)
        cv, vi = ms.encode_canvas(polar, keys, spec, 1, n_dev=offs[1:], canvas=persistent, return_index=True)
        ops.clear_canvas_cells(cv, vi)
    g = torch.cuda.CUDAGraph()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        run()
    t = timeit(g.replay, n=200, warm=20)
    polar = ops.cart_to_polar(cart)
    _, keys = ops.grid_index(polar, offs, 1, spec, want_grid_ind=False)
    V = ops.build_voxel_index(keys, spec, 1).count()
    C_ = ms.reader.out_channels
    nbytes = 30000 * 7 * 4 + V * 4 * 8 + V * C_ * 4 + C_ * spec.grid[0] * spec.grid[1] * 4
    print(f"C2  scatter stage (persistent canvas + sparse clear), {tag} grid {spec.grid[0]} x {spec.grid[1]}: {t * 1e3:.1f} us per 30k-pt sweep (hipGraph), V = {V}, "
          f"{nbytes / 1e6:.1f} MB algorithmic -> {nbytes / t / 1e6:.0f} GB/s ({nbytes / t / 1e6 / 8000:.3f} of 8 TB/s)")

scatter_row("config", synth.NUSC_RANGE, synth.NUSC_VOXEL)
scatter_row("BASELINE synthetic", synth.COARSE_RANGE, synth.COARSE_VOXEL)

# ---- C4 stages
sw = torch.from_numpy(synth.synth_sweep_polar(180000, seed=0, rho_max=74.0)).to(dev)
vg = VoxelGenerator(synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
def vox():
    v, c, n, nv = ops.hard_voxelize(sw, vg.voxel_size, vg.point_cloud_range, 5, 150000)
    return ops.hard_voxel_mean(v, n)
print(f"C4  hard voxelize 180k pts (1152x2048x40, P=5, Vmax=150k) + mean VFE: {timeit(vox):.3f} ms")
pos = waymo_bev_pos()
blks = []
for i in range(2):
    b = SetBlock(in_dim=256, embed_dim_scale=1, num_heads=4, reso=(144, 256), mlp_ratio=4.0, qkv_bias=True, H_sp=144, W_sp=1, H=4, W=8,
                 pos=pos, shift=(i == 1)); synth.load_filled(b, 70 + i); blks.append(b.to(dev).eval())
x = torch.randn((1, 144 * 256, 256), device=dev)
def attn():
    y = x
    for b in blks:
        y = b(y)
    return y
t = timeit(attn)
print(f"C4  2 x SetBlock (36864 tokens x 256), eager launches: {t:.3f} ms  ({123.2 / t:.1f} TFLOP/s on 123.2 GFLOP)")
neck = P.build_neck(dict(type="RPN", layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256], us_layer_strides=[1, 2],
                         us_num_filters=[256, 256], num_input_features=256, logger=logging.getLogger("RPN")))
synth.load_filled(neck, 3); neck = neck.to(dev).eval()
xb = torch.randn((1, 256, 144, 256), device=dev)  # NHWC (B, theta, r, C)
t = timeit(lambda: neck.forward_nhwc(xb))
print(f"C4  RPN of the Waymo config on (256 x 144) x 256: {t:.3f} ms  ({143.14 / t:.1f} TFLOP/s on 143.14 GFLOP)")

# ---- C4 geometry-aware head (H3): E2ESWVoteHead on the RPN output (B, 512, 256, 144)
import numpy as np  # noqa: E402
tasks = [dict(num_class=1, class_names=["VEHICLE"])]
head = P.build_bbox_head(dict(
    type="E2ESWVoteHead", in_channels=512, tasks=tasks, dataset="waymo", weight=2, code_weights=[1.0] * 8, out_size_factor=8,
    common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)}, voxel_shape="cylinder",
    CODER_CONFIG={"code_size": 7, "encode_angle_by_sincos": True},
    GT_PROCESSOR_CONFIG={"max_volumn_space": [75.18, 3.14368, 4.0], "min_volumn_space": [0.3, -3.14368, -2.0], "grid_size": np.array([1152, 2048, 40])},
    HEAD_CONFIG={"kernel_size": 3, "sw_head_version": "votev4", "window_size": 7, "sl_depth": [2], "code_size": 7, "encode_angle_by_sincos": True,
                 "iou_loss": True, "init_bias": -2.19, "num_classes": 1}))
geo = {k: getattr(head, k).clone() for k in ("offset_grid", "xy_offset")}
synth.load_filled(head, 4)
for k, v in geo.items():
    getattr(head, k).data.copy_(v)
head = head.to(dev).eval()
xh = torch.randn((1, 256, 144, 512), device=dev)  # NHWC
t = timeit(lambda: head.forward_nhwc(xh))
print(f"C4  E2ESWVoteHead (256 x 144, 512 ch, 2 Swin blocks of 777 windows x 4 heads): {t:.3f} ms")

# ---- C4 RPN with bf16 convolutions (BASELINE configs[3])
neck.set_compute_dtype("bf16")
ref32 = None
t = timeit(lambda: neck.forward_nhwc(xb))
y16 = neck.forward_nhwc(xb)
neck.set_compute_dtype("f32")
y32 = neck.forward_nhwc(xb)
err = float((y16 - y32).abs().max() / y32.abs().max())
print(f"C4  RPN of the Waymo config, bf16 convs (f32 accumulate): {t:.3f} ms  ({143.14 / t:.1f} TFLOP/s on 143.14 GFLOP); max rel diff to the f32 path {err:.2e}")

# ---- decode + rotated NMS (next-2) on the nuScenes head map (B=1, 128 x 128 x 10 classes)
out = m.forward_points(ops.cart_to_polar(torch.from_numpy(synth.synth_sweep_cart(30000, seed=1)).to(dev)), torch.tensor([0, 30000], dtype=torch.int32, device=dev), 1)
tcfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.1, out_size_factor=4, voxel_size=synth.NUSC_VOXEL,
            pc_range=synth.NUSC_RANGE, nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2))
for k in out:
    out[k] = out[k].contiguous(memory_format=torch.channels_last) if out[k].stride(1) != 1 else out[k]
out["hm"] = out["hm"] * 0 + torch.randn_like(out["hm"]) * 2.0 - 3.0          # random-init weights give no peaks: synthetic logits
t = timeit(lambda: m.bbox_head.predict(dict(metadata=[None]), {"det_preds": [out]}, tcfg), n=20, warm=5)
n_det = m.bbox_head.predict(dict(metadata=[None]), {"det_preds": [out]}, tcfg)[0]["scores"].numel()
print(f"C2  decode + rotated NMS (128 x 128 x 10, pre 1000 / post 83, incl. the one host sync): {t:.3f} ms  ({n_det} boxes)")

# ---- C4 end to end: Waymo PARTNER model, one 180k-point sweep (B = 1): voxelize -> VFE -> sparse backbone -> 2 x SetBlock -> RPN -> E2ESWVoteHead
cfg4 = P.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "waymo", "polar_partner_c4.py"))
m4 = P.build_detector(cfg4.model, train_cfg=cfg4.train_cfg, test_cfg=None)
geo = {k: getattr(m4.bbox_head, k).clone() for k in ("offset_grid", "xy_offset")}
synth.load_filled(m4, 31)
for k, v in geo.items():
    getattr(m4.bbox_head, k).data.copy_(v)
m4 = m4.to(dev).eval()

def frame4():
    voxels, coors, num = vg.generate(sw)
    coords4 = torch.cat([torch.zeros((coors.shape[0], 1), dtype=coors.dtype, device=dev), coors], 1)
    ex = dict(voxels=voxels, coordinates=coords4, num_points=num, num_voxels=[int(voxels.shape[0])], shape=[np.array([1152, 2048, 40])])
    return m4(ex, return_loss=False)

def sparse_only():
    voxels, coors, num = vg.generate(sw)
    coords4 = torch.cat([torch.zeros((coors.shape[0], 1), dtype=coors.dtype, device=dev), coors], 1)
    return m4.backbone.forward_nhwc(m4.reader(voxels, num), coords4, 1, [1152, 2048, 40])

t = timeit(sparse_only)
print(f"C4  voxelize + VFE + SpMiddleResNetFHD, UNIFORM random 180k points (dilates to a nearly dense pyramid): {t:.3f} ms")
sw_uniform = sw
sw = torch.from_numpy(synth.synth_sweep_beams_polar(180000, seed=0)).to(dev)   # 64-beam sweep over ground + obstacles
t = timeit(sparse_only)
print(f"C4  voxelize + VFE + SpMiddleResNetFHD, 64-beam synthetic sweep (surfaces): {t:.3f} ms")
t = timeit(frame4)
print(f"C4  end to end, f32 (B=1, 180k pts): {t:.3f} ms  ({1e3 / t:.1f} frames/s)")
# decode + NMS of the geometry-aware head with the config's test_cfg (pre 4096 / post 500 / IoU 0.7) on lifted logits
_p4 = frame4()
_p4["det_preds"][0]["hm"] = _p4["det_preds"][0]["hm"] + 3.0
t = timeit(lambda: m4.bbox_head.predict(dict(metadata=[None]), _p4, cfg4.test_cfg, device_only=True), n=20, warm=5)
print(f"C4  E2ESWVoteHead decode + rotated NMS (256 x 144 map, pre 4096 / post 500, device only): {t:.3f} ms "
      f"({int(m4.bbox_head.predict(dict(metadata=[None]), _p4, cfg4.test_cfg, device_only=True)['count'][0])} boxes)")
m4.neck.set_compute_dtype("bf16"); m4.bbox_head.set_compute_dtype("bf16")
t = timeit(frame4)
print(f"C4  end to end, bf16 BEV convs (RPN + head branches): {t:.3f} ms  ({1e3 / t:.1f} frames/s)")

# ---- C5 complete: RAW 10-sweep frame -> accumulate -> ... -> decode + NMS, one hipGraph replay per frame
from partner_amd.engine import StreamingFrameEngine
seng = StreamingFrameEngine(m, n_sweeps=10, raw_capacity=310000, test_cfg=tcfg).capture()
frames5 = []
for s in range(4):
    clouds, mats, lags = synth.synth_raw_sweeps(10, 30000, seed=40 + s)
    frames5.append((torch.from_numpy(np.concatenate(clouds, 0)).to(dev),
                    torch.tensor(np.concatenate([[0], np.cumsum([len(c) for c in clouds])]), dtype=torch.int32, device=dev),
                    torch.from_numpy(mats).to(dev), torch.from_numpy(lags).to(dev)))
lat = []
for i in range(220):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = seng.run(*frames5[i % 4]); torch.cuda.synchronize()
    if i >= 20:
        lat.append(1e3 * (time.perf_counter() - t0))
lat = np.sort(np.array(lat))
print(f"C5  RAW 10-sweep frame (300k pts) -> boxes, one hipGraph (accumulate + model + decode/NMS): p50 {lat[len(lat)//2]:.3f} ms  p99 {lat[int(len(lat)*0.99)]:.3f} ms; "
      f"{int(seng.offsets[1])} points kept, {int(o['count'][0])} boxes")

# ---- C4 as a hipGraph: one 180k-point sweep per replay (B = 1), bf16 RPN convolutions
cart4 = torch.from_numpy(synth.synth_sweep_beams_cart(180000, seed=0)).to(dev)
eng4 = FrameEngine(m4, 1, 180000).capture()
lat = []
for i in range(120):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng4.run(cart4); torch.cuda.synchronize()
    if i >= 20:
        lat.append(1e3 * (time.perf_counter() - t0))
lat = np.sort(np.array(lat))
print(f"C4  hipGraph replay per frame (cart points -> head tensors, bf16 BEV convs): p50 {lat[len(lat)//2]:.3f} ms  p99 {lat[int(len(lat)*0.99)]:.3f} ms ({1e3/lat[len(lat)//2]:.1f} frames/s)")

# ---- the same frame on to BOXES inside the graph (E2ESWVoteHead.predict, config test_cfg); classification bias lifted so that the
#      NMS has its full 4096 candidates to work on (random-init weights give no peaks)
with torch.no_grad():
    [mod for mod in m4.bbox_head.cls_head.modules() if isinstance(mod, torch.nn.Conv2d)][-1].bias.add_(4.0)
m4.bbox_head._plan = type(m4.bbox_head._plan)()
eng4b = FrameEngine(m4, 1, 180000, test_cfg=cfg4.test_cfg).capture()
lat = []
for i in range(120):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o4 = eng4b.run(cart4); torch.cuda.synchronize()
    if i >= 20:
        lat.append(1e3 * (time.perf_counter() - t0))
lat = np.sort(np.array(lat))
print(f"C4  hipGraph replay per frame, cart points -> BOXES (bf16 BEV convs, decode + NMS of 4096 candidates in the graph): p50 {lat[len(lat)//2]:.3f} ms  "
      f"p99 {lat[int(len(lat)*0.99)]:.3f} ms; {int(o4['count'][0])} boxes")
