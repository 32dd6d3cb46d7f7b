"""Bitwise run-to-run determinism of the training step and of the conv tiles (run on the GPU box).
usage: python tools/determinism_check.py [train|conv]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def train(reps=40):
    from tests.test_hip_train import _small_train_setup
    golden = lambda name: np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name))
    g, m, ts, tg, pts, gi = _small_train_setup(torch.device("cuda"), golden)
    p0 = ts.ps.flat_p.clone()
    stats0 = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    ref = None
    bad = {}
    for r in range(reps):
        ts.ps.flat_p.copy_(p0)
        m.load_state_dict(stats0, strict=False)
        loss = ts.forward_backward(pts, None, 2, tg, grid_ind=gi).clone()
        cur = {k: v.clone() for k, v in ts.ps.g.items()}
        cur["__loss"] = loss
        for i, b in enumerate(ts.block_out):
            cur[f"__block{i}"] = b.clone()
        if ref is None:
            ref = cur
            continue
        for k in ref:
            if not torch.equal(ref[k], cur[k]):
                bad.setdefault(k, []).append((r, float((ref[k] - cur[k]).abs().max())))
    print("train: tensors that differed run-to-run:", {k: v[:3] for k, v in bad.items()} or "none")


def conv(reps=200):
    from partner_amd import ops
    torch.manual_seed(0)
    for (cin, cout, hw, k, stride) in [(128, 128, 128, 3, 1), (64, 128, 128, 3, 1), (128, 128, 64, 3, 1), (64, 64, 128, 3, 1), (128, 256, 64, 3, 2), (64, 3, 128, 3, 1)]:
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        x = ops.to_nhwc(torch.randn(2, cin, hw, hw, device="cuda"))
        layer = ops.ConvLayer(w, stride=stride, pad=k // 2, act=1)
        ref = layer(x).clone()
        nbad = sum(int(not torch.equal(ref, layer(x))) for _ in range(reps))
        print(f"conv {cin}->{cout} {hw}x{hw} k{k} s{stride} tile={os.environ.get('PN_CONV_TILE', 'auto')}: {nbad}/{reps} differ")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "train"
    {"train": train, "conv": conv}[what]()
