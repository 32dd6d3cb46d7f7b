#!/usr/bin/env python3
"""Headline benchmark: frames/s of the polar voxelize + PFN + BEV backbone + centre head hot
path (BASELINE.json configs[1]: nuScenes polar-pillar cfg, synthetic 30k-point sweeps, forward
only, fp32) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launched bare with N > 1 (no RANK / WORLD_SIZE in the environment) the script starts N child
processes of itself, one per GPU, before anything touches the GPU (the reference launches one
process per GPU too: README.md:64, tools/train.py:103-107), waits for them and exits with their
status; rank 0 prints the line.

One step = one pass of the hot path over one batch (default: 1 sweep) that is already resident
in HBM as Cartesian points: cart->polar (V0), grid indices (V1), bitmap unique-rank, bucketing,
fused PFN + canvas (V4/V5), RPN (B1), CenterHeadSinglePos (H2) -- head tensors out.
Frames are independent, so ranks just process their own frames (weak scaling, no data-path
collective); the timed region is bracketed by barrier + device synchronise and the maximum
over ranks is reported.

Objects in the JSON line:
  value / ms_per_step        the timed region: K hipGraph replays, `--streams` frames in flight (default: 3 or 4, measured)
  single_stream_ms_per_step  the same K replays with ONE frame in flight (latency regime) -- the regime `roofline` is
                             quoted in, so roofline.conv_ms_per_step <= single_stream_ms_per_step
  roofline          the dominant kernel (fp32-MFMA implicit-GEMM convolution): algorithmic FLOPs of every launch / its
                    own execution time (events attached to the dispatch, = rocprofv3's kernel duration), one stream
  roofline_scatter  the scatter stage V0..V5 against the HBM roofline: algorithmic bytes of the variant that is run
                    (persistent canvas + sparse clear: the dense canvas is never moved) / time of the stage's hipGraph
  train_step        BASELINE configs[2]: the DDP training iteration at bs = 4 sweeps per GPU (fwd + loss + bwd + bucketed
                    gradient all-reduce overlapped with backward + clip / wd / Adam)
  cpu_baseline      the CPU oracle (oracle/polar_oracle.py, a port of the reference path checked against reference
                    goldens) timed on this box's host cores, rank 0, N=1 only.
  c4                BASELINE configs[3] (N=1): the Waymo PARTNER detector at bs = 2, f32 and with bf16 BEV convolutions: ms per
                    step, per-stage ms, issued-MFMA roofline fraction per MFMA stage (partner_amd/utils/legs.py)
  c5                BASELINE configs[4] (N=1): 300k-point streaming frames, one hipGraph replay per frame, latency p50 / p99
  roofline_scatter_coarse   SURVEY 8(d)'s secondary row: the scatter stage on the 0.3125 m x 0.05 rad synthetic grid
  ranks             N > 1: what every rank saw (world size after init, device ordinal, PCI bus id, host)

roofline.frac is the matrix work ISSUED (MFMA FLOPs the kernels execute: Winograd launches issue 2/3 resp. 1/2 of the direct
algorithm's FLOPs) / kernel time / 157.3 TFLOP/s, so it cannot exceed 1; the direct algorithm's FLOPs over the same time are
reported next to it as `algorithmic_equiv` (first layer billed dense there, with the pairs it multiplies in `frac`).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from partner_amd.utils import synth  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_TBS = 8.0            # same guide: HBM3E ~8 TB/s

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


def host_cpu_info():
    """(model name, physical cores, logical cpus) of this host from /proc/cpuinfo"""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or logical or (os.cpu_count() or 1)), (logical or (os.cpu_count() or 1))


def cpu_baseline(n_points: int, batch: int, budget_s: float = 30.0, min_frames: int = 20, max_frames: int = 40, warmups: int = 5):
    """time the CPU oracle on the same workload (bounded sample: >= 20 timed frames, <= ~30 s)"""
    from oracle import polar_oracle as O

    cfg = c2_model_cfg()

    class _S:
        def __init__(self, s):
            self.shape = s

    import partner_amd as P
    shapes = {k: _S(tuple(v.shape)) for k, v in P.build_detector(cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 0).items()}
    model_name, phys, logical = host_cpu_info()
    # torch CPU convolutions stop scaling (and collapse) beyond ~16 threads on the GPU box's host
    # (measured r1: 8 thr 0.36 s, 16 thr 0.28 s, 32 thr 0.33 s, 64 thr 0.70 s, 256 thr 34 s per frame)
    threads = min(16, phys)
    torch.set_num_threads(threads)
    times = []
    t_all = time.perf_counter()
    with torch.no_grad():
        for f in range(max_frames + warmups):
            sweeps = [synth.synth_sweep_cart(n_points, seed=1000 + f * batch + b) for b in range(batch)]
            t0 = time.perf_counter()
            polar = [O.cart_to_polar(s) for s in sweeps]
            gi = O.with_batch_index([O.grid_index(p, synth.NUSC_RANGE, synth.NUSC_VOXEL) for p in polar])
            O.pointpillars_forward(sd, cfg, np.concatenate(polar, 0), gi, batch)
            dt = time.perf_counter() - t0
            if f >= warmups:
                times.append(dt)
            if len(times) >= max_frames or (time.perf_counter() - t_all > budget_s and len(times) >= min_frames):
                break
    ts = np.sort(np.array(times))
    fps = batch * len(times) / sum(times)
    return dict(value=round(fps, 4), unit="frames/s", cores=torch.get_num_threads(), kind="port",
                cpu_model=model_name, host_physical_cores=phys, host_logical_cpus=logical,
                p50_s_per_frame=round(float(ts[len(ts) // 2]), 4),
                sample=f"{len(times)} timed frames (+{warmups} warm-up) of the same {n_points}-pt synthetic sweeps, batch {batch}, "
                       f"oracle/polar_oracle.py (numpy + torch CPU fp32, {torch.get_num_threads()} threads: more threads are slower "
                       f"on this host); survey-container cross-check of the actual reference: 1.05-1.3 frames/s on 8 vCPU (SURVEY 6)")


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int) -> int:
    """bare `python bench.py --gpus N`: start N children of this script (fresh interpreters -- nothing is re-exec'd and
    this parent never touches the GPU), one rank per GPU, and return the worst exit status"""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:   # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------ training leg
class TrainLeg:
    """BASELINE configs[2]: the DDP training iteration, bs = `batch` sweeps per GPU, fp32"""

    def __init__(self, model, dev, rank, batch, n_points, total_steps):
        from partner_amd import ops
        from partner_amd.train import PolarPillarTrainStep
        self.ops, self.B, self.N = ops, batch, n_points
        self.ts = PolarPillarTrainStep(model, total_steps=max(100, total_steps))
        self.ts.sync_initial_params()
        pool = 2
        self.frames = []
        for f in range(pool):
            cart = np.concatenate([synth.synth_sweep_cart(n_points, seed=(rank * pool + f) * batch + b) for b in range(batch)], 0)
            self.frames.append(torch.from_numpy(cart).to(dev))
        self.offs = torch.tensor([n_points * b for b in range(batch + 1)], dtype=torch.int32, device=dev)
        # targets as SURVEY 8d prescribes for C3: K = 40 random boxes per frame through the (device) polar target assignment
        gb = torch.zeros((batch, 64, 9), dtype=torch.float32)
        gc = torch.zeros((batch, 64), dtype=torch.int32)
        for b in range(batch):
            boxes, classes = synth.synth_gt_boxes(40, seed=1000 + rank * batch + b)
            gb[b, :40], gc[b, :40] = torch.from_numpy(boxes), torch.from_numpy(classes.astype(np.int32))
        self.tg = ops.assign_heatmap_polar(gb.to(dev), gc.to(dev), torch.full((batch,), 40, dtype=torch.int32, device=dev), 10, 500, [128, 128],
                                           np.float32(synth.NUSC_VOXEL), np.float32(synth.NUSC_RANGE), 4, 0.1, 2)

    def step(self, i):
        polar = self.ops.cart_to_polar(self.frames[i % len(self.frames)])
        return self.ts.step(polar, self.offs, self.B, self.tg)

    def allreduce_ms(self, reps=10):
        """the gradient exchange alone: `reps` bucketed all-reduces of the flat gradient buffer, back to back"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return 0.0
        g = torch.zeros_like(self.ts.ps.flat_g)
        for _ in range(2):
            dist.all_reduce(g)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            for lo, hi in self.ts.buckets:
                dist.all_reduce(g[lo:hi])
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps


def run_train(args, model, dev, rank, world, red_dev, steps, warmup):
    from partner_amd import dist_utils as D
    leg = TrainLeg(model, dev, rank, args.train_batch, args.points, steps + warmup)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    loss0 = None
    for i in range(warmup):
        loss = leg.step(i)
        loss0 = float(loss[0]) if loss0 is None else loss0
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = leg.step(i)
    host_issue_ms = 1e3 * (time.perf_counter() - t0) / steps     # the host's share: time to QUEUE an iteration (no synchronisation in the loop)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, red_dev)
    last_loss = float(loss[0])
    ar = D.max_over_ranks(leg.allreduce_ms(), red_dev)
    # the same iterations with the exchange switched off (timing only; the ranks' parameters diverge from here on, nothing
    # follows): exposed communication = ms_per_iter - ms_per_iter_no_exchange
    no_ex = None
    if world > 1:
        leg.ts.exchange_enabled = False
        for i in range(2):
            leg.step(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(steps):
            leg.step(i)
        barrier()
        no_ex = 1e3 * D.max_over_ranks(time.perf_counter() - t1, red_dev) / steps
        leg.ts.exchange_enabled = True
    B = args.train_batch
    return dict(ms_per_iter=round(1e3 * elapsed / steps, 3), frames_per_s=round(world * steps * B / elapsed, 3), sweeps_per_iter_per_gpu=B,
                iters=steps, warmup=warmup, n_gpus=world, host_issue_ms_per_iter=round(host_issue_ms, 3),
                ms_per_iter_no_exchange=None if no_ex is None else round(no_ex, 3),
                exposed_exchange_ms=None if no_ex is None else round(1e3 * elapsed / steps - no_ex, 3),
                all_reduce_ms=round(ar, 3), all_reduce="flat fp32 gradient buffer in reverse-layer-order buckets, issued during backward "
                                                       f"({leg.ts.ps.total} floats, {len(leg.ts.buckets)} buckets); all_reduce_ms = the exchange alone, back to back, "
                                                       "not overlapped; exposed_exchange_ms = ms_per_iter - ms_per_iter_no_exchange (the same iterations with the "
                                                       "collectives switched off)",
                approx_tflops_per_gpu=round(3 * 150.6e9 * B * steps / elapsed / 1e12, 1),  # fwd 150.6 GFLOP/frame, bwd = dgrad + wgrad
                first_loss=loss0, last_loss=last_loss)


def run_train_partner(args, dev, rank, world, red_dev, steps, warmup):
    """BASELINE configs[3] as a TRAINING iteration: the Waymo PARTNER detector (VoxelNetV3: sparse encoder, SetBlocks, RPN,
    E2ESWVoteHead + set criterion), bs = 2 synthetic 180k-point sweeps per GPU, fp32, gradients all-reduced over RCCL"""
    import partner_amd as P
    from partner_amd import dist_utils as D
    from partner_amd.train_partner import PartnerTrainStep
    from partner_amd.voxel_generator import VoxelGenerator
    cfg_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "waymo", "polar_partner_c4.py")
    w = P.Config.fromfile(cfg_path)
    m = P.build_detector(w.model, train_cfg=w.train_cfg, test_cfg=None)
    geo = {k: getattr(m.bbox_head, k).clone() for k in ("offset_grid", "xy_offset")}
    synth.load_filled(m, base_seed=31)
    for k, v in geo.items():
        getattr(m.bbox_head, k).data.copy_(v)
    m = m.to(dev).train()
    B = 2
    vg = VoxelGenerator(synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
    vs, cs, ns, counts = [], [], [], []
    for b in range(B):
        sw = torch.from_numpy(synth.synth_sweep_beams_polar(180000, seed=rank * B + b)).to(dev)
        voxels, coors, num = vg.generate(sw)
        vs.append(voxels)
        ns.append(num)
        cs.append(torch.cat([torch.full((coors.shape[0], 1), b, dtype=coors.dtype, device=dev), coors], 1))
        counts.append(int(voxels.shape[0]))
    ex = dict(voxels=torch.cat(vs), coordinates=torch.cat(cs), num_points=torch.cat(ns), num_voxels=counts, shape=[np.array([1152, 2048, 40])] * B,
              global_box=torch.from_numpy(synth.synth_vehicle_boxes(B, 40, seed=2 + rank)))
    step = PartnerTrainStep(m, total_steps=max(100, steps + warmup))
    step.sync_initial_params()

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    first = None
    for _ in range(warmup):
        losses = step.step(ex)
        first = float(losses["det_loss"][0]) if first is None else first
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = step.step(ex)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, red_dev)
    return dict(ms_per_iter=round(1e3 * elapsed / steps, 3), frames_per_s=round(world * steps * B / elapsed, 3), sweeps_per_iter_per_gpu=B, iters=steps,
                warmup=warmup, n_gpus=world, parameters=step.ps.total, voxels_per_iter=int(sum(counts)), first_loss=first,
                last_loss=float(losses["det_loss"][0]))


# ------------------------------------------------------------------------------------------------ scatter roofline
def scatter_roofline(model, dev, n_points, spec):
    """V0..V5 alone (cart->polar, grid index, unique-rank, bucket, PFN, canvas write, sparse clear) as one hipGraph,
    timed with HIP events over 200 replays on the current stream"""
    from partner_amd import ops
    cart = torch.from_numpy(synth.synth_sweep_cart(n_points, seed=5)).to(dev)
    offs = torch.tensor([0, n_points], dtype=torch.int32, device=dev)
    persistent, state = model.new_canvas(1, spec, dev), model.new_index_state(1, spec, dev)

    def run():
        return model.scatter_stage(cart, offs, 1, spec, persistent, state)

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
    for _ in range(20):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    c = model.reader.out_channels
    b_in = n_points * 5 * 4                     # Cartesian points read (x, y, z, intensity, dt)
    b_feat = v * c * 4                          # pillar features written (into their canvas cells)
    b_unq = v * 4 * 8                           # SURVEY 8(d) bills the unq table; the fused path keeps int32 keys (v * 4 B) instead
    b_clear = v * c * 4                         # sparse clear of the same cells after the backbone's first layer
    alg = n_points * 7 * 4 + b_unq + b_feat     # SURVEY 8(d) "without canvas" variant: 16.2 MB at V = 28.3k
    pmc = committed_pmc_bytes(("fused_polar_index", "cell_scan", "order_fill", "dynamic_pfn", "clear_frame_cells"))
    return dict(bound="hbm", stage="V0..V5 (fused cart->polar + grid index + cell counts, one look-back scan = unique ranks + point slots, bucket fill, "
                                   "fused PFN + canvas cells, sparse clear): 5 launches",
                variant="persistent canvas + sparse clear: the 134 MB dense canvas is never filled or moved",
                points=n_points, voxels=v, bytes_algorithmic=alg, bytes_algorithmic_with_clear=alg + b_clear,
                bytes_breakdown=dict(points_in=b_in, polar_points=n_points * 7 * 4, unq=b_unq, features=b_feat, clear=b_clear),
                bytes_pmc=pmc, us=round(us, 2), achieved=round(alg / us / 1e6, 4), peak=PEAK_HBM_TBS, unit="TB/s",
                frac=round(alg / us / 1e6 / PEAK_HBM_TBS, 4), frac_with_clear=round((alg + b_clear) / us / 1e6 / PEAK_HBM_TBS, 4),
                measured="HIP events around 200 replays of the stage's own hipGraph on one stream")


def pmc_traffic_path():
    """the newest committed PMC traffic summary (profiles/rN_pmc_traffic.csv, highest N)"""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.csv")):
        m = re.match(r"r(\d+)_pmc_traffic\.csv$", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    return None if best is None else best[1]


PMC_FRAME_KERNEL = "fused_polar_index_kernel"      # one launch per frame in every regime: the frame count of a PMC pass


def committed_pmc_bytes(kernel_prefixes, per="frame"):
    """HBM bytes from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as the gfx950 guide prescribes + WRITE_SIZE, separate
    --pmc passes of this command): summed over the kernels whose name contains one of the prefixes, per frame (per='frame') or averaged
    per launch (per='launch').  None if there is no summary at all or no kernel matches; a summary WITHOUT the per-frame kernel is an
    error (the frame count would be a guess)."""
    import csv
    path = pmc_traffic_path()
    if path is None:
        return None
    rows = list(csv.DictReader(open(path)))
    frames = [int(r["launches"]) for r in rows if PMC_FRAME_KERNEL in r["kernel"]]
    if not frames:
        raise RuntimeError(f"{os.path.basename(path)}: no row for {PMC_FRAME_KERNEL} -- the frame count of the PMC pass is unknown")
    frames = frames[0]
    tot, n = 0.0, 0
    for r in rows:
        if any(p in r["kernel"] for p in kernel_prefixes):
            k = int(r["launches"])
            tot += k * (float(r["fetch_KB_corrected_x2_avg"]) + float(r["WRITE_SIZE_KB_avg"])) * 1024.0
            n += k
    if not n:
        return None
    return round(tot / frames) if per == "frame" else round(tot / n)


# ------------------------------------------------------------------------------------------------ main
def rank_table(rank, world, dev_index=None):
    """what every rank saw after the rendezvous: -> (ranks_seen, [per-rank dict]) on every rank.  ranks_seen is
    dist.get_world_size() AFTER init (1 without a process group): the line is self-verifying when the driver runs N > 1"""
    import torch.distributed as dist
    me = dict(rank=rank, pid=os.getpid(), host=socket.gethostname(), world_size_seen=1, device=dev_index, pci_bus_id=None)
    if dev_index is not None:
        try:
            from partner_amd import hip
            me["pci_bus_id"] = hip.device_info(dev_index).get("pci_bus_id")
        except Exception as e:  # noqa: BLE001 -- the identity of the device is a report, never a reason to lose the line
            me["pci_bus_id"] = f"unavailable: {e}"
    if not (dist.is_available() and dist.is_initialized()):
        return 1, [me]
    me["world_size_seen"] = dist.get_world_size()
    table = [None] * dist.get_world_size()
    dist.all_gather_object(table, me)
    return dist.get_world_size(), sorted(table, key=lambda r: r["rank"])


def dry_run(args, rank, world):
    """launcher / rendezvous / barrier / max-reduce on any backend without a GPU (CPU tests of the N > 1 plumbing)"""
    from partner_amd import dist_utils as D
    D.init(args.backend, None, timeout_s=args.collective_timeout)
    D.barrier()
    seen, ranks = rank_table(rank, world)
    # the frame shard of the inference path (frame i -> rank i mod world, dist_utils.frame_shard) over world x 8 frame ids: every rank
    # reports its own, rank 0 checks that the union covers every frame exactly once
    frame_ids = list(range(world * 8))
    mine = D.frame_shard(frame_ids, rank, world)
    shards = [mine]
    if world > 1:
        shards = [None] * world
        torch.distributed.all_gather_object(shards, mine)
    flat = sorted(f for sh in shards for f in sh)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    # the two timings of the training leg's exchange accounting, on sleeps: same fields as the GPU line
    train = dict(ms_per_iter=round(1e3 * elapsed / args.steps, 3), ms_per_iter_no_exchange=round(1e3 * elapsed / args.steps, 3) if world > 1 else None,
                 exposed_exchange_ms=0.0 if world > 1 else None, n_gpus=world)
    if rank == 0:
        print(json.dumps({"metric": "dry run of the launcher and the rank plumbing (no GPU work, no product path)", "value": round(world * args.steps / elapsed, 3),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "dry-run",
                          "config": {"workload": "none (--dry-run)", "parallelism": f"ranks x{world}", "backend": args.backend},
                          "ranks_seen": seen, "ranks": ranks, "train_step": train,
                          "frame_shard": {"frames": len(frame_ids), "per_rank": [len(sh) for sh in shards], "covered_exactly_once": flat == frame_ids}}), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="sweeps per step per GPU of the inference path")
    ap.add_argument("--train-batch", type=int, default=4, help="sweeps per training iteration per GPU (BASELINE configs[2]: 4)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "train-partner"],
                    help="infer: BASELINE configs[1], the headline metric, with the training iteration as the `train_step` object "
                         "(default); train: configs[2] as the headline line (K timed training iterations); train-partner: the training iteration of "
                         "the Waymo PARTNER detector, configs[3], bs = 2 per GPU")
    ap.add_argument("--points", type=int, default=30000, help="points per sweep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the throughput measurement at the reference config's batch of 4 sweeps per step")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replays")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for a same-GPU dry run)")
    ap.add_argument("--streams", type=int, default=0,
                    help="frames in flight per GPU (one hipGraph engine per HIP stream); 0 (default, r6): FramePipeline captures four engines and MEASURES "
                         "three against four in flight at construction, like the stream assignment and the chain form (r4 fixed 3: 1537 against 1482 "
                         "frames/s over the driver's 20 steps; with the r6 kernels four are 2.5 % faster sustained and 0.5 - 2 % over 20 steps)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise the launcher, the rendezvous and the reductions only")
    ap.add_argument("--collective-timeout", type=float, default=300.0, help="seconds a rank waits in a collective before it gives up (N > 1)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 200-step repeat of the timed loop (`value_sustained`; counter passes of the profiler)")
    ap.add_argument("--no-c4", action="store_true", help="skip the Waymo PARTNER leg (BASELINE configs[3], N = 1 only)")
    ap.add_argument("--no-c5", action="store_true", help="skip the 300k-point streaming leg (BASELINE configs[4], N = 1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)     # before anything touches the GPU

    from partner_amd import dist_utils as D
    rank, local_rank, world = D.env_rank_world()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or run bare and let bench.py start them)", file=sys.stderr)
        return 2
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: there is no CPU fallback for the product path (use --dry-run to test the launcher)", file=sys.stderr)
        return 3
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        print(f"bench.py: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    # RCCL over xGMI: barrier, MAX-reduce of the time and the gradient buckets of the training leg; every wait is bounded
    D.init(args.backend, dev, timeout_s=args.collective_timeout)
    ranks_seen, ranks = rank_table(rank, world, dev.index)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    import partner_amd as P
    from partner_amd import hip, ops
    hip.load()

    model = P.build_detector(c2_model_cfg())
    synth.load_filled(model, base_seed=0)
    model = model.to(dev).eval()
    ops.probe_streams(dev)        # the one-time stream probing happens here, not inside the first forward

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    B, N = args.batch, args.points
    if args.mode == "train-partner":
        del model
        tr = run_train_partner(args, dev, rank, world, red_dev, min(args.steps, 20), min(args.warmup, 3))
        if rank == 0:
            print(json.dumps({
                "metric": "frames/sec DDP training step of the Waymo PARTNER detector (fwd + set criterion + bwd + grad all-reduce + clip/wd/Adam), 180k-pt sweeps (whole job)",
                "value": tr["frames_per_s"], "unit": "frames/s", "n_gpus": world, "steps": tr["iters"], "warmup": tr["warmup"],
                "ms_per_step": tr["ms_per_iter"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "Waymo polar PARTNER cfg (VoxelNetV3: mean VFE -> SpMiddleResNetFHD -> 2 x SetBlock -> RPN -> E2ESWVoteHead), "
                                       "training iteration (BASELINE configs[3])", "points_per_sweep": 180000, "sweeps_per_step_per_gpu": 2,
                           "parallelism": f"dp{world}, flat-gradient all-reduce"},
                "train_step": tr, "ranks_seen": ranks_seen, "ranks": ranks}), flush=True)
        if world > 1:
            D.barrier()
            torch.distributed.destroy_process_group()
        return 0
    if args.mode == "train":
        tr = run_train(args, model, dev, rank, world, red_dev, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps({
                "metric": "frames/sec DDP training step (fwd + loss + bwd + grad all-reduce + clip/wd/Adam), 30k-pt sweeps (whole job)",
                "value": tr["frames_per_s"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": tr["ms_per_iter"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "nuScenes polar-pillar PARTNER cfg training iteration (BASELINE configs[2])", "points_per_sweep": N,
                           "sweeps_per_step_per_gpu": args.train_batch, "parallelism": f"dp{world}, bucketed flat-gradient all-reduce overlapped with backward"},
                "train_step": tr, "ranks_seen": ranks_seen, "ranks": ranks}), flush=True)
        if world > 1:
            D.barrier()
            torch.distributed.destroy_process_group()
        return 0

    pool = 8  # distinct resident frames per rank
    frames = []
    for f in range(pool):
        cart = np.concatenate([synth.synth_sweep_cart(N, seed=(rank * pool + f) * B + b) for b in range(B)], 0)
        frames.append(torch.from_numpy(cart).to(dev))
    offs = torch.tensor([N * b for b in range(B + 1)], dtype=torch.int32, device=dev)
    spec = ops.GridSpec.from_range(synth.NUSC_RANGE, synth.NUSC_VOXEL)

    eager_canvas, eager_state = model.new_canvas(B, spec, dev), model.new_index_state(B, spec, dev)

    def step_eager(i):
        return model.forward_cart(frames[i % pool], offs, B, spec, canvas=eager_canvas, index_state=eager_state)   # V0 .. H2

    engines = []
    stream_tuning = None
    if not args.eager:
        # the multi-frame regime as ONE object: FramePipeline captures the engines and measures their stream assignment at construction
        # (engine.py: an unmeasured assignment can sit 20 % lower; "untuned_ms_per_frame" in the line is that penalty on this box)
        from partner_amd.engine import FrameEngine, FramePipeline, tune_replay_streams
        # (the candidates are timed in bursts of the length this run times, from an idle chip like the timed loop itself: over 20 steps the
        # fill of a four-deep pipeline costs what it gains over 200)
        pipe = FramePipeline(model, B, N, spec, frames_in_flight=(3, 4) if args.streams <= 0 else args.streams, form_rounds=6,
                             form_frames=min(128, max(16, args.steps)))
        engines, stream_tuning = pipe.engines, pipe.tuning
    depth = len(engines) if engines else 1      # frames in flight of the timed loop (measured by FramePipeline when --streams is 0)

    def step(i):
        # consecutive frames go to alternating streams: independent frames overlap on the GPU
        return engines[i % len(engines)].run(frames[i % pool], sync=False) if engines else step_eager(i)

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, red_dev)

    # the same loop over >= 200 steps (the K timed steps above are 13 ms at the default K = 20: this is the sustained figure beside them)
    sustained = None
    if engines and not args.no_sustained:
        ks = max(200, 4 * args.steps)
        barrier()
        t1 = time.perf_counter()
        for i in range(ks):
            step(i)
        barrier()
        es = D.max_over_ranks(time.perf_counter() - t1, red_dev)
        sustained = dict(value=round(world * ks * B / es, 3), unit="frames/s", steps=ks, ms_per_step=round(1e3 * es / ks, 4))

    # the secondary legs run on an auxiliary stream and leave the default stream to the training leg (which picks a second stream that
    # really overlaps with it: ops.concurrent_stream)
    aux_stream = torch.cuda.Stream(device=dev)
    aux_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(aux_stream)

    # the same K frames with ONE frame in flight (one engine, one stream): the regime the per-kernel figures are quoted in
    # (an engine of its own: the engines above are captured with the hint that several frames are in flight, this one has the chip to itself)
    single_ms = None
    if engines:
        lat_engine = engines[0] if len(engines) == 1 else FrameEngine(model, B, N, spec).capture()
        for i in range(20):      # (a freshly captured graph: its first replays carry the upload; 20 = 16 ms, untimed)
            lat_engine.run(frames[i % pool], sync=False)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            lat_engine.run(frames[i % pool], sync=False)
        barrier()
        single_ms = 1e3 * D.max_over_ranks(time.perf_counter() - t1, red_dev) / args.steps
        if lat_engine is not engines[0]:
            del lat_engine

    # the same frames at the reference config's own batch (data.samples_per_gpu = 4, configs/nusc/pp/polarstream_det_n_seg_1_sector.py:199;
    # tools/dist_test.py:118 evaluates with it and times with 1 under --speed_test): four sweeps per graph replay, two replays in flight.
    # The 128 x 128 layers then have enough tiles for the plain F(4,3) form and every launch serves four frames.  Reported next to the
    # headline, which stays at one sweep per step like the reference's speed test; per-frame results are identical (eval-mode BatchNorm /
    # per-sample GroupNorm).
    batched = None
    if engines and B == 1 and not args.no_batched:
        from partner_amd.engine import FrameEngine
        GB, GS = 4, min(2, depth)
        groups = [torch.cat([frames[(GB * f + j) % pool] for j in range(GB)], 0) for f in range(max(1, pool // GB))]
        eng2 = []
        for k in range(GS):
            st = torch.cuda.Stream() if GS > 1 else None
            eng2.append(FrameEngine(model, GB, N, spec, frames_in_flight=GS).capture(stream=st))
        if len(eng2) > 1:
            tune_replay_streams(eng2, groups[0], trials=6, frames=12)
        k2 = max(4, args.steps // 2)
        for i in range(max(2, args.warmup // GB)):
            eng2[i % len(eng2)].run(groups[i % len(groups)], sync=False)
        barrier()
        t1 = time.perf_counter()
        for i in range(k2):
            eng2[i % len(eng2)].run(groups[i % len(groups)], sync=False)
        barrier()
        e2 = D.max_over_ranks(time.perf_counter() - t1, red_dev)
        batched = dict(sweeps_per_step_per_gpu=GB, steps=k2, ms_per_step=round(1e3 * e2 / k2, 4), value=round(world * k2 * GB / e2, 3), unit="frames/s",
                       launch=f"hipGraph replay per {GB} frames (the reference config's samples_per_gpu), {GS} replays in flight")
        eng2.clear()

    # roofline passes: the same K steps launched eagerly on one stream with an event pair attached to every conv DISPATCH
    # (hipExtLaunchKernelGGL start/stop events = the kernel's own execution time; events cannot ride inside a replayed graph).
    # r6: `roofline` describes the kernel FORMS of the regime `value` is timed in -- the launches carry the
    # frames-in-flight hint and the chain form FramePipeline chose, so its dominant kernel is the headline's (conv_wchain3_kernel<2> when
    # F(4,3)xF(4,3) was chosen) --, each launch timed alone on the chip (a kernel's own roofline); the forms a frame takes when it has the
    # chip to itself are the same measurement under `roofline_single_stream`.
    def conv_roofline(hint, chain44_on, full):
        with ops.frames_in_flight(hint), ops.chain44(chain44_on):
            # (eager launches keep the GPU a third busy: after two warm-up frames the first timed ones still ran at the clocks of an idle
            # device and the dominant kernel's fraction moved between 0.63 and 0.67 from run to run; 30 frames = 60 ms settle it)
            for i in range(30):
                step_eager(i)
            prof = ops.enable_conv_profiling()
            barrier()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step_eager(i)
            barrier()
            eager_ms = 1e3 * (time.perf_counter() - t1) / args.steps
            flops, ms, launches, tags = prof.collect(by_tag=True, full=True)
            ops.disable_conv_profiling()
        # frac: the matrix work ISSUED / kernel time / peak.  Winograd launches issue 6 (F(2,3)) or 4.5
        # (F(4,3)) of the direct algorithm's 9 MACs per output; block 0's first layer multiplies its (pillar, tap) pairs only.
        issued_f = sum(t[3] for t in tags.values())
        dense_f = sum(t[4] for t in tags.values())
        issued = issued_f / (ms * 1e-3) / 1e12
        dense = dense_f / (ms * 1e-3) / 1e12
        dom_tag, dom = max(tags.items(), key=lambda kv: kv[1][1])
        dom_tf = dom[3] / (dom[1] * 1e-3) / 1e12
        if dom_tag.startswith("256x256") and "F(4,3)xF(4,3) chain" in dom_tag:
            dom_name, dom_pmc_name, alg = "conv_wchain3_kernel<2>", "conv_wchain3_kernel<2>", "1/4"
        elif dom_tag.startswith("256x256") and "F(2,3)xF(4,3) chain" in dom_tag:
            dom_name, dom_pmc_name, alg = "conv_wchain2_kernel<1,1,2>", "conv_wchain2_kernel<1, 1, 2>", "1/3"
        else:
            dom_name, dom_pmc_name, alg = ("conv_wchain_kernel" if "chain" in dom_tag else "conv_mfma_kernel"), None, "see by_layer"
        traffic = committed_pmc_bytes((dom_pmc_name,), per="launch") if dom_pmc_name else None
        r = dict(bound="mfma", kernel=f"{dom_name} (conv_wchain.hip), fp32 v_mfma_f32_32x32x2_f32: layer {dom_tag}",
                 regime=(f"kernel forms of the headline regime ({hint} frames in flight, chain form chosen by FramePipeline's measurement), each launch "
                         "timed alone on the chip") if hint > 1 else "kernel forms of ONE frame in flight (a frame has the chip to itself)",
                 achieved=round(dom_tf, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(dom_tf / PEAK_F32_MFMA_TFLOPS, 4),
                 traffic=traffic, traffic_source=os.path.basename(pmc_traffic_path() or "none"),
                 dominant_us=round(1e3 * dom[1] / dom[2], 2), dominant_launches_per_step=round(dom[2] / args.steps, 2),
                 dominant_gflop_issued_per_launch=round(dom[3] / dom[2] / 1e9, 3), dominant_share_of_conv_time=round(dom[1] / ms, 3),
                 all_conv_frac=round(issued / PEAK_F32_MFMA_TFLOPS, 4), conv_ms_per_step=round(ms / args.steps, 4),
                 launches_per_step=round(launches / args.steps, 2))
        if hint > 1:
            r["frac_in_flight"] = round(issued_f / elapsed / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)      # (both loops ran args.steps steps)
        if not full:
            return r
        layers = {t: dict(launches_per_step=round(n / args.steps, 2), us=round(1e3 * m / n, 2), tflops_issued=round(iss / (m * 1e-3) / 1e12, 1),
                          frac=round(iss / (m * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 3), tflops_algorithmic_equiv=round(den / (m * 1e-3) / 1e12, 1))
                  for t, (f, m, n, iss, den) in sorted(tags.items(), key=lambda kv: -kv[1][1])}
        r.update(counts="the DOMINANT kernel of these forms (most time per frame): the FLOPs it ISSUES to the matrix pipes per launch "
                        f"({alg} of the direct algorithm's 9 MACs per output: chained F(4,3)xF(4,3) issues 36 products per 16 outputs, F(2,3)xF(4,3) "
                        "1/3, F(4,3) 1/2) / its average launch duration (start/stop events on its own dispatches); frac_in_flight = issued FLOPs of "
                        "ALL conv launches per frame / ms_per_step (the timed headline loop itself) / peak; all_conv_frac = the r1-r3 definition",
                 traffic_unit="HBM bytes per launch of the dominant kernel (offline PMC passes of this command: FETCH_SIZE x 2 + WRITE_SIZE); algorithmic: "
                              "planes in + planes out = 2 x 1.5 x pixels x C x 4 B = 100.7 MB at 256^2 x 128 (+ 1.2 MB of weights)",
                 traffic_all_conv=committed_pmc_bytes(("conv_mfma_kernel", "conv_wino", "conv_wchain", "conv_small_n", "conv_multi", "pair_gemm",
                                                       "pair_reduce", "conv_s2", "pillar_tile"), per="launch"),
                 algorithmic_equiv=dict(tflops=round(dense, 3), over_peak=round(dense / PEAK_F32_MFMA_TFLOPS, 4),
                                        note="the direct dense algorithm's FLOPs (every layer 2*pixels*Cout*Cin*KH*KW, first layer included) over the "
                                             "kernel time of all conv launches: what the FLOP-reducing forms buy; not a roofline fraction, may exceed 1"),
                 measured="start/stop events attached to every conv dispatch (hipExtLaunchKernelGGL) over the same K steps "
                          "launched eagerly on one stream: per-kernel execution time as in a rocprofv3 kernel trace",
                 eager_ms_per_step=round(eager_ms, 4), by_layer=layers)
        return r

    roofline = roofline_single = None
    if not args.no_roofline_events:
        k_hint = depth
        roofline = conv_roofline(k_hint, pipe.chain44 if engines else True, full=True)
        if k_hint > 1:
            roofline_single = conv_roofline(1, True, full=False)
            roofline_single["paired_with"] = "single_stream_ms_per_step"

    # every rank runs the stage measurement (rank 0's is reported): no rank waits in a collective while another measures alone
    scatter = coarse = scatter300 = None
    if not args.no_roofline_events and B == 1:
        scatter = scatter_roofline(model, dev, N, spec)
        try:      # the same stage on a C5-sized frame (300k points in one sweep: ~180k pillars; bytes from N and V, no PMC row for this size)
            scatter300 = scatter_roofline(model, dev, 300000, spec)
            scatter300["bytes_pmc"] = None
        except Exception as e:  # noqa: BLE001
            scatter300 = dict(error=f"{type(e).__name__}: {e}")
        from partner_amd.utils import legs
        try:
            coarse = legs.coarse_scatter_row(c2_model_cfg(), dev, N)
        except Exception as e:  # noqa: BLE001 -- a secondary row never costs the headline line
            coarse = dict(error=f"{type(e).__name__}: {e}")

    # BASELINE configs[4] and configs[3] on this GPU (N = 1 only: secondary objects of the default line)
    c5 = c4 = None
    if world == 1 and B == 1 and not args.eager:
        from partner_amd.utils import legs
        if not args.no_c5:
            try:
                tcfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.1, out_size_factor=4,
                            voxel_size=synth.NUSC_VOXEL, pc_range=synth.NUSC_RANGE,
                            nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2))
                c5 = legs.c5_leg(model, dev, tcfg)
            except Exception as e:  # noqa: BLE001
                c5 = dict(error=f"{type(e).__name__}: {e}")

    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    train = None
    if not args.no_train_leg:
        engines.clear()        # free the graphs' private pools before the training iteration allocates its activations
        torch.cuda.empty_cache()
        train = run_train(args, model, dev, rank, world, red_dev, steps=min(args.steps, 20), warmup=5)

    torch.cuda.set_stream(aux_stream)
    if world == 1 and B == 1 and not args.eager and not args.no_c4:
        from partner_amd.utils import legs
        engines.clear()
        torch.cuda.empty_cache()
        try:
            c4 = legs.c4_leg(dev)
        except Exception as e:  # noqa: BLE001
            c4 = dict(error=f"{type(e).__name__}: {e}")

    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    if rank == 0:
        fps = world * args.steps * B / elapsed

        def get(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d

        form = get(stream_tuning, "chain_form") or {}
        # the secondary figures in one flat object, printed FIRST and LAST in the line (records that keep only the head or the tail of
        # the line keep them) and, as one short string, inside `config` (records that keep the contract's keys only)
        summary = dict(
            value=round(fps, 3), value_sustained=None if sustained is None else sustained["value"],
            single_stream_ms_per_step=None if single_ms is None else round(single_ms, 4),
            chain_form=form.get("chosen"), chain_form_ms_per_frame=form.get("candidates"),
            frames_in_flight=depth, depth_ms_per_frame=get(stream_tuning, "depth", "candidates"),
            roofline_kernel=None if roofline is None else roofline["kernel"].split(" ")[0], roofline_frac=get(roofline, "frac"),
            roofline_frac_in_flight=get(roofline, "frac_in_flight"), roofline_single_stream_frac=get(roofline_single, "frac"),
            scatter_us=get(scatter, "us"), scatter_300k_us=get(scatter300, "us"),
            train_ms_per_iter=get(train, "ms_per_iter"), batched_frames_per_s=get(batched, "value"),
            c4_f32_ms_per_step=get(c4, "f32", "ms_per_step"), c4_f32_one_graph_p50_ms=get(c4, "one_graph_bs2", "p50_ms"),
            c4_f32_sparse_encoder_ms=get(c4, "f32", "stages", "sparse_encoder", "ms"), c4_f32_setblocks_x2_ms=get(c4, "f32", "stages", "setblocks_x2", "ms"),
            c4_f32_rpn_ms=get(c4, "f32", "stages", "rpn", "ms"), c4_f32_head_ms=get(c4, "f32", "stages", "e2e_swv_head", "ms"),
            c4_f32_dense_as_run_ms=get(c4, "f32", "dense_stages_as_run", "ms"),
            c4_bf16_ms_per_step=get(c4, "option_bf16_bev_convs", "ms_per_step"),
            c4_bf16_one_graph_p50_ms=get(c4, "option_bf16_bev_convs", "one_graph_bs2", "p50_ms"),
            c4_bf16_rpn_ms=get(c4, "option_bf16_bev_convs", "stages", "rpn", "ms"), c4_bf16_head_ms=get(c4, "option_bf16_bev_convs", "stages", "e2e_swv_head", "ms"),
            c5_p50_ms=get(c5, "p50_ms"), c5_p99_ms=get(c5, "p99_ms"))

        def f(x, nd=2):
            return "-" if x is None else (f"{x:.{nd}f}" if isinstance(x, float) else str(x))

        cands = form.get("candidates") or {}
        secondary = (f"sust {f(summary['value_sustained'], 0)}|1s {f(summary['single_stream_ms_per_step'], 3)}|tr {f(summary['train_ms_per_iter'])}|"
                     f"c4 {f(summary['c4_f32_one_graph_p50_ms'])}/{f(summary['c4_bf16_one_graph_p50_ms'])}|c5 {f(summary['c5_p50_ms'], 3)}|"
                     f"f44 {f(cands.get('F(4,3)xF(4,3)'), 3)} f24 {f(cands.get('F(2,3)xF(4,3)'), 3)}")
        line = {
            "metric": "frames/sec polar voxelize+PFN+BEV+head, 30k-pt sweep (whole job)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "summary": summary,
            "config": {"workload": "nuScenes polar-pillar PARTNER cfg (DynamicPFNet -> DynamicPPScatter -> RPN -> "
                                   "CenterHeadSinglePos), grid 512x512x1, forward only (BASELINE configs[1])",
                       "points_per_sweep": N, "sweeps_per_step_per_gpu": B, "parallelism": f"frame-replicas x{world}",
                       "launch": "eager" if args.eager else f"hipGraph replay per frame, {depth} frame(s) in flight on separate HIP streams" + (" (depth measured: 3 against 4)" if args.streams <= 0 else ""),
                       "chain_form": form.get("chosen"), "secondary": secondary, "device": hip.device_info(dev.index or 0)},
            "value_sustained": None if sustained is None else sustained["value"], "sustained": sustained,
            "single_stream_ms_per_step": None if single_ms is None else round(single_ms, 4),
            "replay_stream_tuning": stream_tuning,
            "roofline": roofline, "roofline_single_stream": roofline_single,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, B)
            line["gpu_over_cpu"] = round(fps / line["cpu_baseline"]["value"], 1)
        line.update({"batched": batched, "roofline_scatter": scatter, "roofline_scatter_300k": scatter300, "roofline_scatter_coarse": coarse,
                     "train_step": train, "c4": c4, "c5": c5, "ranks_seen": ranks_seen, "ranks": ranks, "summary_tail": summary})
        print(json.dumps(line), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
