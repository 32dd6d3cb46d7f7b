#!/usr/bin/env python3
"""Headline benchmark: frames/s of the polar voxelize + PFN + BEV backbone + centre head hot
path (BASELINE.json configs[1]: nuScenes polar-pillar cfg, synthetic 30k-point sweeps, forward
only, fp32) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch (default: 1 sweep) that is already resident
in HBM as Cartesian points: cart->polar (V0), grid indices (V1), bitmap unique-rank, bucketing,
fused PFN + canvas (V4/V5), RPN (B1), CenterHeadSinglePos (H2) -- head tensors out.
Frames are independent, so ranks just process their own frames (weak scaling, no data-path
collective); the timed region is bracketed by barrier + device synchronise and the maximum
over ranks is reported.

Extra objects in the JSON line:
  roofline     -- the dominant kernel (fp32-MFMA implicit-GEMM convolution): algorithmic FLOPs
                  of every launch / its HIP-event duration, both summed over the timed region.
  cpu_baseline -- the CPU oracle (oracle/polar_oracle.py, a port of the reference path checked
                  against reference goldens) timed on this box's host cores, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from partner_amd.utils import synth  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"

TASKS = [dict(num_class=10, class_names=["car", "truck", "construction_vehicle", "bus", "trailer", "barrier",
                                         "motorcycle", "bicycle", "pedestrian", "traffic_cone"])]


def c2_model_cfg():
    """nuScenes polar-pillar model (values of SURVEY.md Appendix A.1)."""
    rng_, vs = list(synth.NUSC_RANGE), list(synth.NUSC_VOXEL)
    vg = dict(range=rng_, voxel_size=vs, max_points_in_voxel=20, max_voxel_num=[30000, 60000], voxel_shape="cylinder",
              return_density=True, dynamic=True, nsectors=1)
    return dict(
        type="PointPillars", pretrained=None,
        reader=dict(type="DynamicPFNet", num_filters=[64, 128], num_input_features=7, voxel_shape="cylinder",
                    xyz_cluster=True, raz_cluster=True, xy_center=True, ra_center=True, voxel_size=vs, pc_range=rng_),
        backbone=dict(type="DynamicPPScatter", ds_factor=1),
        neck=dict(type="RPN", layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[128, 128, 256],
                  us_layer_strides=[0.5, 1, 2], us_num_filters=[128, 128, 128], num_input_features=128,
                  logger=logging.getLogger("RPN")),
        bbox_head=dict(type="CenterHeadSinglePos", in_channels=384, tasks=TASKS, dataset="nuscenes", weight=0.5,
                       code_weights=[1.5, 1.5, 1.0, 1.0, 1.0, 1.0, 0.5, 0.5, 1.0, 1.0],
                       common_heads={"reg": (2, 2), "rot_vel": (2, 2), "height": (1, 2), "dim": (3, 2)},
                       voxel_shape="cylinder", voxel_generator=vg),
        seg_head=None, part_head=None)


def cpu_baseline(n_points: int, batch: int, budget_s: float = 20.0, max_frames: int = 10):
    """time the CPU oracle on the same workload (bounded sample)"""
    from oracle import polar_oracle as O

    cfg = c2_model_cfg()

    class _S:
        def __init__(self, s):
            self.shape = s

    import partner_amd as P
    shapes = {k: _S(tuple(v.shape)) for k, v in P.build_detector(cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 0).items()}
    # torch CPU convolutions stop scaling (and collapse) beyond ~16 threads on the GPU box's host
    # (measured: 8 thr 0.36 s, 16 thr 0.28 s, 32 thr 0.33 s, 64 thr 0.70 s, 256 thr 34 s per frame)
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    times = []
    t_all = time.perf_counter()
    with torch.no_grad():
        for f in range(max_frames + 1):
            sweeps = [synth.synth_sweep_cart(n_points, seed=1000 + f * batch + b) for b in range(batch)]
            t0 = time.perf_counter()
            polar = [O.cart_to_polar(s) for s in sweeps]
            gi = O.with_batch_index([O.grid_index(p, synth.NUSC_RANGE, synth.NUSC_VOXEL) for p in polar])
            O.pointpillars_forward(sd, cfg, np.concatenate(polar, 0), gi, batch)
            dt = time.perf_counter() - t0
            if f > 0:  # first frame is warm-up
                times.append(dt)
            if time.perf_counter() - t_all > budget_s and len(times) >= 2:
                break
    fps = batch * len(times) / sum(times)
    return dict(value=round(fps, 4), unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(times)} timed frames (+1 warm-up) of the same {n_points}-pt synthetic sweeps, batch {batch}, "
                       f"oracle/polar_oracle.py (numpy + torch CPU fp32, {torch.get_num_threads()} threads)")


def train_mode(args, model, dev, rank, world, red_dev):
    """BASELINE configs[2]: the DDP training iteration, bs = --batch sweeps per GPU, fp32"""
    from partner_amd import dist_utils as D
    from partner_amd import ops
    from partner_amd.train import PolarPillarTrainStep
    B, N = args.batch, args.points
    ts = PolarPillarTrainStep(model, total_steps=max(100, args.steps + args.warmup))
    ts.sync_initial_params()
    pool = 2
    frames = []
    for f in range(pool):
        cart = np.concatenate([synth.synth_sweep_cart(N, seed=(rank * pool + f) * B + b) for b in range(B)], 0)
        frames.append(torch.from_numpy(cart).to(dev))
    offs = torch.tensor([N * b for b in range(B + 1)], dtype=torch.int32, device=dev)
    # targets as SURVEY 8d prescribes for C3: K = 40 random boxes per frame through the (device) polar target assignment
    gb = torch.zeros((B, 64, 9), dtype=torch.float32)
    gc = torch.zeros((B, 64), dtype=torch.int32)
    for b in range(B):
        boxes, classes = synth.synth_gt_boxes(40, seed=1000 + rank * B + b)
        gb[b, :40], gc[b, :40] = torch.from_numpy(boxes), torch.from_numpy(classes.astype(np.int32))
    tg = ops.assign_heatmap_polar(gb.to(dev), gc.to(dev), torch.full((B,), 40, dtype=torch.int32, device=dev), 10, 500, [128, 128],
                                  np.float32(synth.NUSC_VOXEL), np.float32(synth.NUSC_RANGE), 4, 0.1, 2)

    def step(i):
        polar = ops.cart_to_polar(frames[i % pool])
        return ts.step(polar, offs, B, tg)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    loss0 = None
    for i in range(args.warmup):
        loss = step(i)
        loss0 = float(loss[0]) if loss0 is None else loss0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, red_dev)
    if rank == 0:
        fps = world * args.steps * B / elapsed
        # algorithmic FLOPs of one iteration: forward 150.6 GFLOP/frame, backward = data + weight gradient
        tf = 3 * 150.6e9 * B * args.steps / elapsed / 1e12
        print(json.dumps({
            "metric": "frames/sec DDP training step (fwd + loss + bwd + grad all-reduce + clip/wd/Adam), 30k-pt sweeps (whole job)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "nuScenes polar-pillar PARTNER cfg training iteration (BASELINE configs[2])", "points_per_sweep": N,
                       "sweeps_per_step_per_gpu": B, "parallelism": f"dp{world}, one flat-gradient all-reduce per step",
                       "flat_gradient_floats": ts.ps.total},
            "approx_tflops_per_gpu": round(tf, 1), "first_loss": loss0, "last_loss": float(loss[0]),
        }), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="sweeps per step per GPU (default 1; 4 in --mode train)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer: BASELINE configs[1], the headline metric (default); train: configs[2], one DDP training "
                         "iteration (forward, loss, backward, flat-gradient all-reduce, clip + wd + Adam) at bs=4/GPU")
    ap.add_argument("--points", type=int, default=30000, help="points per sweep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replays")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for a same-GPU dry run)")
    ap.add_argument("--streams", type=int, default=4, help="frames in flight per GPU (one hipGraph engine per HIP stream)")
    args = ap.parse_args()

    from partner_amd import dist_utils as D
    rank, local_rank, world = D.env_rank_world()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the product path"
    ndev = torch.cuda.device_count()
    assert args.backend != "nccl" or local_rank < ndev, f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible"
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    D.init(args.backend, dev)  # RCCL over xGMI; only the barrier and the MAX-reduce of the time use it
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    import partner_amd as P
    from partner_amd import hip, ops
    hip.load()

    model = P.build_detector(c2_model_cfg())
    synth.load_filled(model, base_seed=0)
    model = model.to(dev).eval()

    if args.batch is None:
        args.batch = 4 if args.mode == "train" else 1
    B, N = args.batch, args.points
    if args.mode == "train":
        return train_mode(args, model, dev, rank, world, red_dev)
    pool = 8  # distinct resident frames per rank
    frames = []
    for f in range(pool):
        cart = np.concatenate([synth.synth_sweep_cart(N, seed=(rank * pool + f) * B + b) for b in range(B)], 0)
        frames.append(torch.from_numpy(cart).to(dev))
    offs = torch.tensor([N * b for b in range(B + 1)], dtype=torch.int32, device=dev)
    spec = ops.GridSpec.from_range(synth.NUSC_RANGE, synth.NUSC_VOXEL)

    def step_eager(i):
        polar = ops.cart_to_polar(frames[i % pool])                 # V0
        return model.forward_points(polar, offs, B, spec)           # V1 .. H2

    engines = []
    if not args.eager:
        from partner_amd.engine import FrameEngine
        for k in range(max(1, args.streams)):
            st = torch.cuda.Stream() if args.streams > 1 else None
            engines.append(FrameEngine(model, B, N, spec).capture(stream=st))

    def step(i):
        # consecutive frames go to alternating streams: independent frames overlap on the GPU
        return engines[i % len(engines)].run(frames[i % pool]) if engines else step_eager(i)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, red_dev)

    # roofline pass: HIP events cannot be recorded inside a replayed graph, so the same K steps are
    # run again eagerly with an event pair around every conv launch (same kernels, same inputs)
    prof, eager_ms = None, None
    if not args.no_roofline_events:
        for i in range(2):
            step_eager(i)
        prof = ops.enable_conv_profiling()
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step_eager(i)
        barrier()
        eager_ms = 1e3 * (time.perf_counter() - t1) / args.steps

    def committed_traffic():
        """HBM bytes per conv launch from the committed rocprofv3 PMC passes (profiles/r1_final_pmc_traffic.csv: FETCH_SIZE doubled as
        the gfx950 guide prescribes + WRITE_SIZE, separate --pmc passes of this same command with --streams 1); None if absent"""
        import csv
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_final_pmc_traffic.csv")
        if not os.path.exists(path):
            return None
        tot, n = 0.0, 0
        for r in csv.DictReader(open(path)):
            if r["kernel"].startswith("conv_mfma_kernel"):
                k = int(r["launches"])
                tot += k * (float(r["fetch_KB_corrected_x2_avg"]) + float(r["WRITE_SIZE_KB_avg"])) * 1024.0
                n += k
        return round(tot / n) if n else None

    roofline = None
    if prof is not None:
        flops, ms, launches = prof.collect()
        ops.disable_conv_profiling()
        ach = flops / (ms * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel="conv_mfma_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit GEMM)",
                        achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        traffic=committed_traffic(), traffic_unit="HBM bytes per conv launch (offline PMC passes, profiles/r1_final_pmc_traffic.csv)",
                        launches=launches, flops_per_launch=round(flops / launches), avg_launch_us=round(1e3 * ms / launches, 2),
                        conv_ms_per_step=round(ms / args.steps, 4),
                        measured="HIP events around every conv launch over the same K steps replayed eagerly "
                                 "(events cannot be recorded inside the hipGraph of the timed region)",
                        eager_ms_per_step=round(eager_ms, 4))

    if rank == 0:
        fps = world * args.steps * B / elapsed
        line = {
            "metric": "frames/sec polar voxelize+PFN+BEV+head, 30k-pt sweep (whole job)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "nuScenes polar-pillar PARTNER cfg (DynamicPFNet -> DynamicPPScatter -> RPN -> "
                                   "CenterHeadSinglePos), grid 512x512x1, forward only (BASELINE configs[1])",
                       "points_per_sweep": N, "sweeps_per_step_per_gpu": B, "parallelism": f"frame-replicas x{world}",
                       "launch": "eager" if args.eager else f"hipGraph replay per frame, {len(engines)} frame(s) in flight on separate HIP streams"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, B)
            line["gpu_over_cpu"] = round(fps / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
