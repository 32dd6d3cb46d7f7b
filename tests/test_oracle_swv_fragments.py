"""oracle/polar_oracle.py's Swin-stage pieces against the fragments of the reference's sw2votev4_util.py that execute
(tests/golden/swv_fragments.npz, written by make_golden.py::gen_swv_fragments from the reference): window partition / reverse, MLP,
PatchEmbed and SwinTransformerBlock.forward around a stand-in attention -- the masked uniform average, which the oracle's REAL
attention reproduces when its weights are q = k = 0, v = x, proj = identity, vote / position MLPs zero.  The fixture holds the
reference's OUTPUTS; inputs and weights are the name-keyed seeded arrays the generator used (synth.seeded_normal / fill_state_dict)."""
import numpy as np
import torch

from oracle import polar_oracle as O
from partner_amd.utils import synth


def frag_input(g, name, *shape):
    return torch.from_numpy(synth.seeded_normal("swv_frag." + name, shape, int(g["seed"])))


def filled(shapes, seed):
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: synth.Shape(*s) for k, s in shapes.items()}, seed).items()}


def mlp_shapes(C, hid, p=""):
    return {p + "fc1.weight": (hid, C), p + "fc1.bias": (hid,), p + "fc2.weight": (C, hid), p + "fc2.bias": (C,)}


def block_weights(C, seed):
    """the fixture block's LayerNorm / MLP weights (reference names) + attention weights that make WindowAttention softmax(mask) @ x"""
    shapes = {"norm1.weight": (C,), "norm1.bias": (C,), "norm2.weight": (C,), "norm2.bias": (C,)}
    shapes.update(mlp_shapes(C, C, "mlp."))
    return identity_attention_weights(filled(shapes, seed), "attn.", C)


def identity_attention_weights(sd, prefix, C, heads=4):
    qkv = torch.zeros((3 * C, C))
    qkv[2 * C:] = torch.eye(C)
    sd.update({prefix + "qkv.weight": qkv, prefix + "qkv.bias": torch.zeros(3 * C), prefix + "proj.weight": torch.eye(C), prefix + "proj.bias": torch.zeros(C),
               prefix + "tau": torch.ones((1, heads, 1, 1)),
               prefix + "rpe.0.weight": torch.zeros((16, 2, 1, 1)), prefix + "rpe.0.bias": torch.zeros(16),
               prefix + "rpe.2.weight": torch.zeros((heads, 16, 1, 1)), prefix + "rpe.2.bias": torch.zeros(heads),
               prefix + "vote_mlp.0.weight": torch.zeros((16, 3, 1)), prefix + "vote_mlp.0.bias": torch.zeros(16),
               prefix + "vote_mlp.2.weight": torch.zeros((C, 16, 1)), prefix + "vote_mlp.2.bias": torch.zeros(C)})
    return sd


def test_window_partition_and_reverse(golden):
    g = golden("swv_fragments.npz")
    B, H, W, C, ws, Hp, Wp = (int(v) for v in g["dims"])
    x = frag_input(g, "part_in", B, Hp, Wp, 8)
    win = O._window_partition(x, ws)
    assert torch.equal(win, torch.from_numpy(g["part_out"]))
    assert torch.equal(O._window_reverse(win, ws, Hp, Wp), x)


def test_mlp_and_patch_embed(golden):
    g = golden("swv_fragments.npz")
    B, H, W, C, ws, Hp, Wp = (int(v) for v in g["dims"])
    seed = int(g["seed"])
    with torch.no_grad():
        y = O.swv_mlp(filled(mlp_shapes(C, C), seed), "", frag_input(g, "mlp_in", 40, C))
        t = O.swv_patch_embed(filled({"proj.weight": (C, 2 * C, 1, 1), "proj.bias": (C,), "norm.weight": (C,), "norm.bias": (C,)}, seed), "",
                              frag_input(g, "pe_in", B, 2 * C, H, W))          # (B, HW, C)
    np.testing.assert_allclose(y.numpy(), g["mlp_out"], rtol=1e-5, atol=2e-6)
    ref = torch.from_numpy(g["pe_out"]).flatten(2).transpose(1, 2)
    np.testing.assert_allclose(t.numpy(), ref.numpy(), rtol=1e-5, atol=5e-6)


def test_swin_block_plumbing_matches_the_reference_block(golden):
    """norm1 / pad / shift / partition / reverse / crop / residual / norm2 + MLP of oracle.swv_swin_block == the reference's
    SwinTransformerBlock.forward; 9 x 11 tokens pad to 14 x 14 (two windows each way), shift 0 and 3"""
    g = golden("swv_fragments.npz")
    B, H, W, C, ws, Hp, Wp = (int(v) for v in g["dims"])
    seed = int(g["seed"])
    x, pos, vote = frag_input(g, "blk_x", B, H * W, C), frag_input(g, "blk_pos", B, H * W, 2), frag_input(g, "blk_vote", B, H * W, 3)
    for shift in (0, ws // 2):
        sd = block_weights(C, seed + 1 + shift)
        with torch.no_grad():
            y = O.swv_swin_block(sd, "", x, H, W, pos, vote, ws, shift, heads=4)
        np.testing.assert_allclose(y.numpy(), g[f"blk_y_shift{shift}"], rtol=2e-5, atol=2e-5)
        if shift:   # the mask and the shift matter: the unshifted block gives something else
            assert float(np.abs(O.swv_swin_block(sd, "", x, H, W, pos, vote, ws, 0, heads=4).numpy() - g[f"blk_y_shift{shift}"]).max()) > 1e-2
