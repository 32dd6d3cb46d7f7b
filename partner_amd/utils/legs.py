"""Secondary legs of bench.py: the measurements of the PARTNER-specific path that ride in the default bench line.

 c4_leg   BASELINE configs[3]: the Waymo polar PARTNER detector (VoxelNetV3 of voxelnet.py:171-301: mean VFE -> sparse 3-D
          encoder (scn.py:97-192) -> 2 x SetBlock (set_transformer.py:118-166) -> RPN (rpn.py:23-159) -> E2ESWVoteHead
          (e2e_swv_head.py:22-201)), bs = 2 synthetic 180k-point 64-beam sweeps, f32 and with bf16 BEV convolutions:
          end-to-end ms per step, per-stage ms, and for the MFMA stages the matrix work ISSUED / time / peak.
 c5_leg   BASELINE configs[4]: nuScenes 10-sweep (300k raw points) streaming frames, one hipGraph replay per frame
          (accumulate -> voxelize -> PFN -> RPN -> head -> decode + rotated NMS), latency p50 / p99 over >= 200 replays.
 coarse_scatter_row  SURVEY 8(d)'s secondary scatter row: the 0.3125 m x 0.05 rad synthetic grid of BASELINE.json's metric.

Everything here runs the product path only (HIP kernels through the C ABI); nothing imports the oracle."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import synth

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: ~2.5 PF dense bf16
PEAK_HBM_TBS = 8.0

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _events():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def time_ms(fn, reps: int, warm: int) -> float:
    """HIP events on the current stream (the stream every kernel of the library is launched on) around `reps` eager calls"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = _events()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def graph_time_ms(fn, reps: int, warm: int):
    """device time of `fn` as ONE hipGraph replay (no host launch gaps: the key-point chains of the SetBlock are a dozen 5 us kernels,
    which an eager Python loop cannot issue fast enough) -> (ms, "hipGraph replay"); stages that synchronise with the host cannot be
    captured and fall back to eager launches -> (ms, "eager launches")"""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(2, warm)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return time_ms(g.replay, reps, 2), "hipGraph replay"
    except Exception:  # noqa: BLE001 -- capture refused (host sync inside the stage): measure the eager launches
        torch.cuda.synchronize()
        return time_ms(fn, reps, warm), "eager launches"


def mfma_account(fn, reps: int = 3):
    """run `fn` with an event pair on every convolution / GEMM dispatch: -> dict(kernel_ms, gflop_algorithmic, gflop_issued,
    launches) per call of fn, bf16 and f32 launches separated"""
    from .. import ops
    fn()
    prof = ops.enable_conv_profiling()
    torch.cuda.synchronize()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    flops, ms, launches, tags = prof.collect(by_tag=True, full=True)
    ops.disable_conv_profiling()
    acc = {"f32": [0.0, 0.0, 0.0, 0], "bf16": [0.0, 0.0, 0.0, 0]}
    for t, (f, m, n, iss, den) in tags.items():
        a = acc["bf16" if (t or "").endswith("bf16") else "f32"]
        a[0] += den
        a[1] += iss
        a[2] += m
        a[3] += n
    out = {}
    for k, (fa, fi, m, n) in acc.items():
        if n:
            out[k] = dict(kernel_ms=m / reps, gflop_algorithmic=fa / reps / 1e9, gflop_issued=fi / reps / 1e9, launches=n // reps)
    top = sorted(tags.items(), key=lambda kv: -kv[1][1])[:6]
    out["top"] = {str(t): dict(launches=n // reps, us=round(1e3 * m / n, 1), tflops_issued=round(iss / (m * 1e-3) / 1e12, 1))
                  for t, (f, m, n, iss, den) in top}
    return out


def _stage(timed, acct=None, survey_gflop=None):
    """one stage row: device ms (events around a hipGraph replay of the stage, or around its eager launches when it cannot be captured)
    + the matrix work its MFMA kernels issued"""
    ms, how = timed if isinstance(timed, tuple) else (timed, "eager launches")
    row = dict(ms=round(ms, 4), timed_as=how)
    if acct:
        issued = sum(v["gflop_issued"] for k, v in acct.items() if k in ("f32", "bf16"))
        alg = sum(v["gflop_algorithmic"] for k, v in acct.items() if k in ("f32", "bf16"))
        kms = sum(v["kernel_ms"] for k, v in acct.items() if k in ("f32", "bf16"))
        bf = acct.get("bf16", {}).get("gflop_issued", 0.0)
        # peak of the mix: time the issued work would take at the dense peaks of the element types it ran in
        t_peak_ms = (issued - bf) / PEAK_F32_MFMA_TFLOPS + bf / PEAK_BF16_MFMA_TFLOPS          # GFLOP / (TFLOP/s) = ms
        row.update(gflop_issued=round(issued, 2), gflop_algorithmic_equiv=round(alg, 2), mfma_kernel_ms=round(kms, 4),
                   mfma_launches=sum(v["launches"] for k, v in acct.items() if k in ("f32", "bf16")),
                   tflops_issued=round(issued / ms, 2), frac=round(t_peak_ms / ms, 4),
                   frac_of="issued MFMA FLOPs / stage wall time / dense peak of the element type (f32 157.3, bf16 2500 TFLOP/s)",
                   bf16_share_of_issued=round(bf / issued, 3) if issued else 0.0, top_kernels=acct.get("top"))
        if survey_gflop is not None:
            row["survey_gflop"] = survey_gflop
    return row


def sparse_account(backbone, fn):
    """matrix work of one pass of the sparse encoder (``fn`` runs it): issued = what its kernels put on the MFMA (grouped form: 32 rows x the
    union mask of every live group, sparse_group.hip; the 16-channel level runs on the VALU and is counted by its pairs), useful = the
    existing (site, tap) pairs only.  Reads device counts back: call outside timed regions."""
    from ..sparse_backbone import SpMiddleResNetFHD
    rec, orig = [], SpMiddleResNetFHD._conv

    def spy(feats, n_rows, nbr, count, cap, layer, act, residual=None, groups=None):
        rec.append((nbr, count, layer, groups))
        return orig(feats, n_rows, nbr, count, cap, layer, act, residual, groups)

    SpMiddleResNetFHD._conv = staticmethod(spy)
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        SpMiddleResNetFHD._conv = staticmethod(orig)
    issued = useful = 0.0
    for nbr, count, layer, groups in rec:
        n = int(count.item())
        f = 2.0 * layer["cin"] * layer["cout"]
        pairs = float((nbr[:n] >= 0).sum())
        useful += pairs * f
        if groups is not None and layer["cout"] >= 32:
            gm = groups[1][:(n + 31) // 32].to(torch.int64) & 0xffffffff
            bits = sum(((gm >> t) & 1) for t in range(layer["taps"]))
            issued += float(bits.sum()) * 32 * f
        else:
            issued += pairs * f
    return dict(gflop_issued=issued / 1e9, gflop_useful=useful / 1e9, convolutions=len(rec))


def build_waymo_partner(dev):
    import partner_amd as P
    cfg = P.Config.fromfile(os.path.join(ROOT, "configs", "waymo", "polar_partner_c4.py"))
    m = P.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=None)
    geo = {k: getattr(m.bbox_head, k).clone() for k in ("offset_grid", "xy_offset")}
    synth.load_filled(m, base_seed=31)
    for k, v in geo.items():
        getattr(m.bbox_head, k).data.copy_(v)
    return m.to(dev).eval(), cfg


def c4_leg(dev, batch: int = 2, points: int = 180000, reps: int = 10, warm: int = 3, rank: int = 0):
    """BASELINE configs[3] on one GPU: bs = `batch` sweeps through the reference's example dict (hard voxels)"""
    from ..voxel_generator import VoxelGenerator
    m, cfg = build_waymo_partner(dev)
    vg = VoxelGenerator(synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
    sweeps = [torch.from_numpy(synth.synth_sweep_beams_polar(points, seed=rank * batch + b)).to(dev) for b in range(batch)]
    grid = [1152, 2048, 40]

    def example():
        vs, cs, ns, nv = [], [], [], []
        for b, sw in enumerate(sweeps):
            voxels, coors, num = vg.generate(sw)[:3]
            vs.append(voxels)
            ns.append(num)
            nv.append(int(voxels.shape[0]))
            cs.append(torch.cat([torch.full((coors.shape[0], 1), b, dtype=coors.dtype, device=dev), coors], 1))
        return dict(voxels=torch.cat(vs), coordinates=torch.cat(cs), num_points=torch.cat(ns), num_voxels=nv, shape=[np.array(grid)] * batch)

    def frame():
        return m(example(), return_loss=False)

    ex = example()
    feats = m.reader(ex["voxels"], ex["num_points"])
    x_sp = m.backbone.forward_nhwc(feats, ex["coordinates"], batch, grid)
    x_at = m.realign_nhwc(x_sp)
    x_rpn = m.neck.forward_nhwc(x_at)

    def st_vox():
        e = example()
        return m.reader(e["voxels"], e["num_points"])

    st_sparse = lambda: m.backbone.forward_nhwc(feats, ex["coordinates"], batch, grid)   # noqa: E731
    st_attn = lambda: m.realign_nhwc(x_sp)                                                # noqa: E731
    st_rpn = lambda: m.neck.forward_nhwc(x_at)                                            # noqa: E731
    st_head = lambda: m.bbox_head.forward_nhwc(x_rpn)                                     # noqa: E731

    out = dict(workload="Waymo polar PARTNER cfg (VoxelNetV3: hard voxels P=5 / Vmax=150k on 1152x2048x40 -> mean VFE -> SpMiddleResNetFHD -> "
                        "2 x SetBlock on 144x256 tokens -> RPN -> E2ESWVoteHead), forward (BASELINE configs[3])",
               points_per_sweep=points, sweeps_per_step=batch, voxels_per_step=int(sum(ex["num_voxels"])), data="synthetic 64-beam sweeps",
               launch="ms_per_step: eager launches of the whole step (the example-dict route reads the voxel counts on the host), HIP events around "
                      "%d steps after %d warm-ups; stages: one hipGraph replay each where the stage is capturable (`timed_as`)" % (reps, warm))
    f32 = dict()
    t = time_ms(frame, reps, warm)
    f32["ms_per_step"], f32["frames_per_s"] = round(t, 4), round(1e3 * batch / t, 2)
    stages = dict()
    stages["voxelize_vfe"] = _stage(time_ms(st_vox, reps, warm))         # ends in a host read of the voxel count (the example dict carries python ints)
    stages["sparse_encoder"] = _stage(graph_time_ms(st_sparse, reps, warm))
    try:      # the stage's matrix work: issued (groups x their tap unions) and useful (existing pairs), as fractions of the f32 MFMA peak
        sa = sparse_account(m.backbone, st_sparse)
        ms_sp = stages["sparse_encoder"]["ms"]
        stages["sparse_encoder"].update(gflop_issued=round(sa["gflop_issued"], 2), gflop_useful=round(sa["gflop_useful"], 2), convolutions=sa["convolutions"],
                                        tflops_issued=round(sa["gflop_issued"] / ms_sp, 2), frac=round(sa["gflop_issued"] / ms_sp / PEAK_F32_MFMA_TFLOPS, 4),
                                        frac_useful=round(sa["gflop_useful"] / ms_sp / PEAK_F32_MFMA_TFLOPS, 4),
                                        frac_of="issued (resp. existing-pair) MFMA FLOPs of the 21 sparse convolutions / stage wall time (index builds, neighbour "
                                                "tables and group sorts included) / 157.3 TFLOP/s")
    except Exception as e:   # noqa: BLE001 -- an accounting row must not take the line down
        stages["sparse_encoder"]["account_error"] = repr(e)[:200]
    stages["setblocks_x2"] = _stage(graph_time_ms(st_attn, reps, warm), mfma_account(st_attn), survey_gflop=123.2 * batch)
    stages["rpn"] = _stage(graph_time_ms(st_rpn, reps, warm), mfma_account(st_rpn), survey_gflop=143.14 * batch)
    stages["e2e_swv_head"] = _stage(graph_time_ms(st_head, reps, warm), mfma_account(st_head), survey_gflop=290.0 * batch)
    f32["stages"] = stages
    f32["stage_sum_ms"] = round(sum(s["ms"] for s in stages.values()), 4)
    # r6: as the detector runs the three dense stages of a batch (VoxelNetV3.dense_stages_nhwc): per sample on two streams -- the stage rows
    # above time each stage's batch launches alone on one stream
    if batch > 1:
        ms_d, how = graph_time_ms(lambda: m.dense_stages_nhwc(x_sp), reps, warm)
        f32["dense_stages_as_run"] = dict(ms=round(ms_d, 4), timed_as=how, what="2 x SetBlock -> RPN -> head of the batch, per sample on two streams "
                                          "(branches of one hipGraph); the same three stages one after the other over the batch: the sum of their rows")
    out["f32"] = f32

    # the same detector as ONE hipGraph replay per frame from Cartesian points (VoxelNetV3.forward_points: voxel / site counts never leave the
    # device, buffers sized by capacity; several sweeps per frame: every sample voxelized on its own, the lists joined on the device)
    from ..engine import FrameEngine
    for gb in sorted({1, batch}):
        key = "one_graph_bs%d" % gb
        try:
            cart = torch.cat([torch.from_numpy(synth.synth_sweep_beams_cart(points, seed=rank * batch + b)).to(dev) for b in range(gb)])
            eng = FrameEngine(m, gb, points).capture()
            lat = []
            for i in range(70):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.run(cart)
                torch.cuda.synchronize()
                if i >= 10:
                    lat.append(1e3 * (time.perf_counter() - t0))
            lat.sort()
            p50 = lat[len(lat) // 2]
            out[key] = dict(p50_ms=round(p50, 4), p99_ms=round(lat[int(len(lat) * 0.99)], 4), frames_per_s_at_p50=round(1e3 * gb / p50, 2), replays=len(lat),
                            measured="host clock around (hipGraphLaunch + device synchronise), %d sweep(s) of %d points per replay, f32, points -> head tensors" % (gb, points))
            del eng
        except Exception as e:   # noqa: BLE001 -- a secondary figure must not take the line down
            out[key] = dict(error=repr(e)[:200])

    # throughput with TWO such frames in flight (two engines of bs = `batch` on their own streams, streams picked by measurement)
    try:
        from ..engine import tune_replay_streams
        cart = torch.cat([torch.from_numpy(synth.synth_sweep_beams_cart(points, seed=rank * batch + b)).to(dev) for b in range(batch)])
        engs = [FrameEngine(m, batch, points, frames_in_flight=2).capture(stream=torch.cuda.Stream()) for _ in range(2)]
        tune_replay_streams(engs, cart, trials=4, frames=6)
        for i in range(4):
            engs[i % 2].run(cart, sync=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nrun = 24
        for i in range(nrun):
            engs[i % 2].run(cart, sync=False)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out["two_graphs_in_flight_bs%d" % batch] = dict(frames_per_s=round(nrun * batch / el, 2), ms_per_step=round(1e3 * el / nrun, 4), steps=nrun,
                                                         measured="two hipGraph engines replaying alternately on two streams, host clock over %d replays" % nrun)
        del engs
    except Exception as e:   # noqa: BLE001
        out["two_graphs_in_flight_bs%d" % batch] = dict(error=repr(e)[:200])

    m.set_compute_dtype("bf16")
    b16 = dict()
    t = time_ms(frame, reps, warm)
    b16["ms_per_step"], b16["frames_per_s"] = round(t, 4), round(1e3 * batch / t, 2)
    x_rpn16 = m.neck.forward_nhwc(x_at)
    st_head16 = lambda: m.bbox_head.forward_nhwc(x_rpn16)                                 # noqa: E731
    b16["stages"] = dict(setblocks_x2=_stage(graph_time_ms(st_attn, reps, warm), mfma_account(st_attn), survey_gflop=123.2 * batch),
                         rpn=_stage(graph_time_ms(st_rpn, reps, warm), mfma_account(st_rpn), survey_gflop=143.14 * batch),
                         e2e_swv_head=_stage(graph_time_ms(st_head16, reps, warm), mfma_account(st_head16), survey_gflop=290.0 * batch))
    try:
        cart = torch.cat([torch.from_numpy(synth.synth_sweep_beams_cart(points, seed=rank * batch + b)).to(dev) for b in range(batch)])
        eng = FrameEngine(m, batch, points).capture()
        lat = []
        for i in range(50):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(cart)
            torch.cuda.synchronize()
            if i >= 10:
                lat.append(1e3 * (time.perf_counter() - t0))
        lat.sort()
        b16["one_graph_bs%d" % batch] = dict(p50_ms=round(lat[len(lat) // 2], 4), p99_ms=round(lat[int(len(lat) * 0.99)], 4), replays=len(lat))
        del eng
    except Exception as e:   # noqa: BLE001
        b16["one_graph_bs%d" % batch] = dict(error=repr(e)[:200])
    b16["note"] = ("VoxelNetV3.set_compute_dtype('bf16') (BASELINE configs[3]): the dense BEV stages on the bf16 matrix pipe -- RPN and head convolutions on "
                   "csrc/conv_bf16.hip (r5: LDS-DMA implicit GEMM + rows form on v_mfma_f32_16x16x32_bf16), the token GEMMs of the SetBlocks and of the "
                   "head's Swin stage on pn_linear_bf16 -- f32 accumulation, f32 LayerNorm / attention cores / residual streams / outputs; the sparse "
                   "encoder stays f32.  Stated tolerance against the oracle: tests/test_hip_swv.py::test_bf16_bev_stage_against_the_oracle.  The f32 "
                   "object above is the parity path.")
    out["option_bf16_bev_convs"] = b16
    m.set_compute_dtype("f32")
    return out


def c5_leg(model, dev, test_cfg, replays: int = 220, warm: int = 20, n_sweeps: int = 10, points_per_sweep: int = 30000):
    """BASELINE configs[4]: RAW 10-sweep frames (300k points) -> boxes, one hipGraph replay per frame, host-observed latency
    (copy of the frame's raw buffers + replay + synchronise) p50 / p99; plus the same from already accumulated points"""
    from ..engine import FrameEngine, StreamingFrameEngine
    total = n_sweeps * points_per_sweep
    seng = StreamingFrameEngine(model, n_sweeps=n_sweeps, raw_capacity=total + 10000, test_cfg=test_cfg).capture()
    frames = []
    for s in range(4):
        clouds, mats, lags = synth.synth_raw_sweeps(n_sweeps, points_per_sweep, seed=40 + s)
        frames.append((torch.from_numpy(np.concatenate(clouds, 0)).to(dev),
                       torch.tensor(np.concatenate([[0], np.cumsum([len(c) for c in clouds])]), dtype=torch.int32, device=dev),
                       torch.from_numpy(mats).to(dev), torch.from_numpy(lags).to(dev)))

    def lat(run):
        ts = []
        for i in range(replays + warm):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o = run(i)
            torch.cuda.synchronize()
            if i >= warm:
                ts.append(1e3 * (time.perf_counter() - t0))
        ts = np.sort(np.array(ts))
        return ts, o

    ts, o = lat(lambda i: seng.run(*frames[i % 4]))
    out = dict(workload="nuScenes polar-pillar cfg, 10-sweep frames (%d raw points), raw sweeps -> accumulate -> V0..H2 -> decode + rotated NMS "
                        "-> boxes, ONE hipGraph replay per frame, one frame in flight (BASELINE configs[4])" % total,
               replays=len(ts), p50_ms=round(float(ts[len(ts) // 2]), 4), p99_ms=round(float(ts[int(len(ts) * 0.99)]), 4),
               mean_ms=round(float(ts.mean()), 4), frames_per_s_at_p50=round(1e3 / float(ts[len(ts) // 2]), 1),
               points_kept=int(seng.offsets[1]), boxes=int(o["count"][0]),
               measured="host clock around (input copy + hipGraphLaunch + device synchronise), per frame")
    del seng
    eng = FrameEngine(model, 1, total).capture()
    acc = [torch.from_numpy(synth.synth_sweep_cart(total, seed=s, n_sweeps=n_sweeps)).to(dev) for s in range(4)]
    ts, _ = lat(lambda i: eng.run(acc[i % 4]))
    out["from_accumulated_points"] = dict(p50_ms=round(float(ts[len(ts) // 2]), 4), p99_ms=round(float(ts[int(len(ts) * 0.99)]), 4), replays=len(ts),
                                          outputs="head tensors (no decode)")
    return out


def coarse_scatter_row(model_cfg, dev, n_points: int = 30000, reps: int = 200):
    """SURVEY 8(d) secondary row: the scatter stage V0..V5 on BASELINE.json's synthetic 0.3125 m x 0.05 rad grid (160 x 126 cells)"""
    import copy

    import partner_amd as P
    from .. import ops
    cfg = copy.deepcopy(model_cfg)
    rng_, vs = list(synth.COARSE_RANGE), list(synth.COARSE_VOXEL)
    cfg["reader"].update(voxel_size=vs, pc_range=rng_)
    ms = P.build_detector(cfg)
    synth.load_filled(ms, 0)
    ms = ms.to(dev).eval()
    spec = ops.GridSpec.from_range(rng_, vs)
    cart = torch.from_numpy(synth.synth_sweep_cart(n_points, seed=5)).to(dev)
    offs = torch.tensor([0, n_points], dtype=torch.int32, device=dev)
    persistent, state = ms.new_canvas(1, spec, dev), ms.new_index_state(1, spec, dev)

    def run():
        return ms.scatter_stage(cart, offs, 1, spec, persistent, state)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            vi = run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    v = vi.count()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    us = 1e3 * time_ms(g.replay, reps, 20)
    c = ms.reader.out_channels
    alg = n_points * 7 * 4 + v * 4 * 8 + v * c * 4
    return dict(grid=[int(spec.grid[0]), int(spec.grid[1])], voxel_size=vs[:2], points=n_points, voxels=int(v), points_per_voxel=round(n_points / max(v, 1), 2),
                bytes_algorithmic=int(alg), bytes_algorithmic_with_clear=int(alg + v * c * 4), us=round(us, 2),
                achieved=round(alg / us / 1e6, 4), peak=PEAK_HBM_TBS, unit="TB/s", frac=round(alg / us / 1e6 / PEAK_HBM_TBS, 4),
                frac_with_clear=round((alg + v * c * 4) / us / 1e6 / PEAK_HBM_TBS, 4),
                note="BASELINE.json's synthetic grid (no reference config uses it, SURVEY F8): ~%d points per pillar" % round(n_points / max(v, 1)))
