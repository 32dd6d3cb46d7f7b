"""Full-size training iterations of the Waymo PARTNER config (bs = 2, two synthetic 180k-point sweeps) for profiling:
python tools/train_partner_profile.py [--steps N]   (rocprofv3 --kernel-trace --stats -- python3 tools/train_partner_profile.py)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_example(dev, batch=2):
    from partner_amd.utils import synth
    from partner_amd.voxel_generator import VoxelGenerator
    vg = VoxelGenerator(synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
    vs, cs, ns, counts = [], [], [], []
    for b in range(batch):
        sw = torch.from_numpy(synth.synth_sweep_beams_polar(180000, seed=b)).to(dev)
        voxels, coors, num = vg.generate(sw)
        vs.append(voxels)
        ns.append(num)
        cs.append(torch.cat([torch.full((coors.shape[0], 1), b, dtype=coors.dtype, device=dev), coors], 1))
        counts.append(int(voxels.shape[0]))
    gbox = synth.synth_vehicle_boxes(batch, 40, seed=2)
    return dict(voxels=torch.cat(vs), coordinates=torch.cat(cs), num_points=torch.cat(ns), num_voxels=counts,
                shape=[np.array([1152, 2048, 40])] * batch, global_box=torch.from_numpy(gbox))


def build_model(dev):
    import partner_amd as P
    from partner_amd.utils import synth
    cfg_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "waymo", "polar_partner_c4.py")
    w = P.Config.fromfile(cfg_path)
    m = P.build_detector(w.model, train_cfg=w.train_cfg, test_cfg=None)
    geo = {k: getattr(m.bbox_head, k).clone() for k in ("offset_grid", "xy_offset")}
    synth.load_filled(m, base_seed=31)
    for k, v in geo.items():
        getattr(m.bbox_head, k).data.copy_(v)
    return m.to(dev).train()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    from partner_amd import hip
    from partner_amd.train_partner import PartnerTrainStep
    hip.load()
    dev = torch.device("cuda:0")
    m = build_model(dev)
    ex = build_example(dev, a.batch)
    step = PartnerTrainStep(m, total_steps=1000)
    step.step(ex)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step.step(ex)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print({"ms_per_iter": round(ms, 2), "batch": a.batch, "det_loss": float(losses["det_loss"][0])})


if __name__ == "__main__":
    main()
