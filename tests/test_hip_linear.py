"""pn_linear_f32 (csrc/linear.hip: the nn.Linear of SetBlock / Mlp, set_transformer.py:37-53, and of the Swin stage of E2ESWVoteHead)
against a float64 reference of the same op: y = act(x @ W^T + b) (+ residual).  Every tile form, ragged rows / columns / K, the
two-phase launches (whole rounds of 128 x 128 tiles + the rest in smaller tiles) and the K-split small-M form.  Tolerance 2e-6 of the
output's range: the kernel accumulates in fp32 FMA chains (exact products), the GELU is the exact-erf form."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from partner_amd import hip
    hip.load()
    return torch.device("cuda:0")


def reference(x, w, b, act, res):
    y = x.double() @ w.double().T
    if b is not None:
        y = y + b.double()
    if act == "gelu":
        y = torch.nn.functional.gelu(y)
    elif act == "relu":
        y = torch.relu(y)
    if res is not None:
        y = y + res.double()
    return y


def run_case(dev, m, k, n, act, bias, res, form, seed):
    from partner_amd import hip, ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((m, k), generator=g)
    w = torch.randn((n, k), generator=g) / np.sqrt(k)
    b = torch.randn((n,), generator=g) if bias else None
    r = torch.randn((m, n), generator=g) if res else None
    layer = ops.GemmLayer(w.to(dev), None if b is None else b.to(dev))
    assert layer.linear
    hip.call("pn_linear_set_tile", form)
    try:
        y = layer(x.to(dev), act={"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "gelu": ops.ACT_GELU}[act], residual=None if r is None else r.to(dev))
    finally:
        hip.call("pn_linear_set_tile", 0)
    ref = reference(x, w, b, act, r)
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, (m, k, n, act, bias, res, form, err)


@pytest.mark.parametrize("form", [0, 22, 21, 12, 11, 1])
def test_linear_forms_ragged_shapes(dev, form):
    cases = [(300, 256, 256, "none", True, True), (129, 36, 20, "relu", True, False), (1000, 260, 100, "gelu", True, True),
             (64, 32, 64, "none", False, False), (2048, 1024, 256, "none", True, True), (517, 64, 132, "gelu", False, True)]
    for i, (m, k, n, act, bias, res) in enumerate(cases):
        run_case(dev, m, k, n, act, bias, res, form, seed=10 * form + i)


def test_linear_two_phase_plans(dev):
    """automatic plans on the token counts of the Waymo BEV map: 36 864 x 256 -> 256 is one round of 128 x 128 tiles + a rest in
    64 x 64 tiles, -> 512 whole rounds + half-size tiles; every row of both phases is checked"""
    for (m, k, n, act) in [(36864, 256, 256, "none"), (36864, 256, 512, "none"), (36864 + 77, 256, 256, "gelu"), (40000, 128, 384, "relu")]:
        run_case(dev, m, k, n, act, True, True, 0, seed=m + n)


def test_linear_matches_the_convolution_route_bitwise_on_plain_sums(dev):
    """pn_linear_f32 adds in the same fp32 FMA order as the r2 route (1x1 convolution on conv_mfma_kernel) whatever tiles it picks:
    identical bits where the epilogue is plain"""
    from partner_amd import ops
    g = torch.Generator().manual_seed(5)
    x, w, b = torch.randn((16384, 256), generator=g).to(dev), (torch.randn((512, 256), generator=g) / 16).to(dev), torch.randn((512,), generator=g).to(dev)
    new = ops.GemmLayer(w, b)
    ops.R.linear = False
    try:
        old = ops.GemmLayer(w, b)
    finally:
        ops.R.linear = True
    assert new.linear and not old.linear
    assert torch.equal(new(x), old(x))


def test_linear_ksplit_entry(dev):
    """pn_linear_ksplit_f32 (GemmLayer(ksplit=True)): the key-point chain shapes and ragged ones against float64; on more than one K chunk,
    with and without bias / GELU / residual"""
    from partner_amd import ops
    g = torch.Generator().manual_seed(9)
    for (m, k, n, act, bias, res) in [(1024, 256, 256, "none", True, True), (2048, 256, 1024, "gelu", True, False), (2048, 1024, 256, "none", True, True),
                                      (1024, 256, 768, "none", True, False), (77, 520, 36, "relu", False, True), (4096, 64, 512, "gelu", True, True)]:
        x = torch.randn((m, k), generator=g)
        w = torch.randn((n, k), generator=g) / np.sqrt(k)
        b = torch.randn((n,), generator=g) if bias else None
        r = torch.randn((m, n), generator=g) if res else None
        layer = ops.GemmLayer(w.to(dev), None if b is None else b.to(dev), ksplit=True)
        assert layer.entry == "pn_linear_ksplit_f32"
        y = layer(x.to(dev), act={"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "gelu": ops.ACT_GELU}[act], residual=None if r is None else r.to(dev))
        ref = reference(x, w, b, act, r)
        err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (m, k, n, act, err)


def test_linear_ksplit_never_reads_past_the_packed_weights(dev):
    """r3 advisor finding: for K not a multiple of 256 the K-split form addressed weight rows past the packed buffer through the scalar
    offset (outside the descriptor's range check).  The packed weights sit between NaN blocks here; the result must be finite and right.
    Also: with no activation a NaN in x reaches the output (the epilogue no longer clamps with fmaxf(v, -inf))"""
    from partner_amd import hip, ops
    lib = hip.load()
    g = torch.Generator().manual_seed(3)
    for (m, k, n) in [(64, 64, 32), (96, 128, 64), (33, 384, 36), (40, 520, 128)]:
        x = torch.randn((m, k), generator=g).to(dev)
        w = (torch.randn((n, k), generator=g) / np.sqrt(k)).to(dev)
        nf = lib.pn_linear_packed_weight_floats(n, k)
        guard = 4 * nf + 4096
        arena = torch.full((guard + nf + guard,), float("nan"), dtype=torch.float32, device=dev)
        packed = arena[guard:guard + nf]
        hip.call("pn_pack_linear_weight_f32", w.data_ptr(), n, k, packed.data_ptr(), hip.stream())
        y = torch.empty((m, n), dtype=torch.float32, device=dev)
        hip.call("pn_linear_ksplit_f32", x.data_ptr(), m, k, k, packed.data_ptr(), n, None, ops.ACT_NONE, None, n, y.data_ptr(), n, hip.stream())
        ref = x.double() @ w.double().t()
        assert torch.isfinite(y).all(), (m, k, n)
        assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-6, (m, k, n)
        x[3, 5] = float("nan")
        hip.call("pn_linear_ksplit_f32", x.data_ptr(), m, k, k, packed.data_ptr(), n, None, ops.ACT_NONE, None, n, y.data_ptr(), n, hip.stream())
        assert torch.isnan(y[3]).all() and torch.isfinite(y[4]).all()
        hip.call("pn_linear_f32", x.data_ptr(), m, k, k, packed.data_ptr(), n, None, ops.ACT_NONE, None, n, y.data_ptr(), n, hip.stream())
        assert torch.isnan(y[3]).all() and torch.isfinite(y[4]).all()


def test_linear_rejects_bad_arguments(dev):
    from partner_amd import hip
    lib = hip.load()
    x = torch.zeros((8, 6), device=dev)
    assert lib.pn_linear_f32(x.data_ptr(), 8, 6, 6, x.data_ptr(), 8, None, 0, None, 0, x.data_ptr(), 8, None) == -1      # k % 4
    assert "multiples of 4" in hip.last_error()
    assert lib.pn_linear_set_tile(13) == -1


@pytest.mark.gpu
def test_first_ksplit_call_of_a_process_uses_the_ksplit_form():
    """r3 regression: in a FRESH process the very first pn_linear_ksplit_f32 call must add in the same order as every later one (the pin
    of the tile form used to be initialised lazily by the tiled path only: the first K-split call ran a tiled form and the first frame of
    a process picked other key points than all later frames)"""
    import subprocess
    import sys
    code = (
        "import torch\n"
        "from partner_amd import hip, ops\n"
        "hip.load(); dev = torch.device('cuda:0')\n"
        "g = torch.Generator().manual_seed(0)\n"
        "x = torch.randn((1024, 256), generator=g).to(dev); w = (torch.randn((256, 256), generator=g) * 0.05).to(dev)\n"
        "layer = ops.GemmLayer(w, None, ksplit=True)\n"
        "a = layer(x).clone(); b = layer(x).clone(); c = layer(x).clone()\n"
        "assert torch.equal(a, b) and torch.equal(b, c), float((a - b).abs().max())\n"
        "print('ok')\n")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("case", [(300, 256, 256, "none", True, True, False), (1000, 256, 1024, "gelu", True, False, True), (4097, 1024, 256, "none", True, True, False),
                                  (145, 64, 48, "relu", False, False, False), (2304, 512, 256, "none", True, False, False), (576, 256, 768, "none", True, False, True)],
                         ids=str)
def test_linear_bf16(dev, case):
    """pn_linear_bf16 (csrc/conv_bf16.hip: the bf16 option of the token GEMMs): bf16 operands are exact in f32, so the float64 product of the
    bf16-rounded x and W is the exact reference of the bf16-in / f32-accumulate kernel up to summation order: 2e-5 of the output's range
    for the f32 output (bias, exact-erf GELU and the residual are applied in f32), one more bf16 rounding (2^-8 relative) for the bf16 output.
    Row counts that are not a multiple of the 144 / 128-row tiles, n not a multiple of the 128-column tile."""
    from partner_amd import ops
    m, k, n, act, bias, res, out16 = case
    g = torch.Generator().manual_seed(sum(case[:3]))
    x = torch.randn((m, k), generator=g)
    w = torch.randn((n, k), generator=g) / np.sqrt(k)
    b = torch.randn((n,), generator=g) if bias else None
    r = torch.randn((m, n), generator=g) if res else None
    layer = ops.GemmLayer(w.to(dev), None if b is None else b.to(dev))
    x16 = ops.to_bf16(x.to(dev))
    a = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "gelu": ops.ACT_GELU}[act]
    ref = reference(x.bfloat16().float(), w.bfloat16().float(), b, act, r)
    y = layer(x16, act=a, residual=None if r is None else r.to(dev))
    assert y.dtype == torch.float32
    assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-5
    if out16:
        y16 = layer(x16, act=a, out_bf16=True)
        assert y16.dtype == torch.bfloat16
        assert ((y16.float().double().cpu() - ref).abs() <= ref.abs() * 2.0 ** -8 + 2e-5 * ref.abs().max()).all()
    # the f32 form of the same layer still runs and is a different arithmetic
    y32 = layer(x.to(dev), act=a, residual=None if r is None else r.to(dev))
    assert not torch.equal(y32, y)


def test_layernorm_bf16_copy(dev):
    """pn_layernorm_bf16out_f32: the f32 rows are those of pn_layernorm_f32 bit for bit, the copy is their round-to-nearest-even bf16"""
    from partner_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn((777, 256), generator=g) * 3 + 0.5).to(dev)
    gamma, beta = torch.randn(256, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
    ref, cm = ops.layernorm(x, gamma, beta, 1e-5, want_chan_mean=True)
    out, cm2, o16 = ops.layernorm(x, gamma, beta, 1e-5, want_chan_mean=True, bf16_copy=True)
    assert torch.equal(out, ref) and torch.equal(cm, cm2) and torch.equal(o16, ref.bfloat16())
    only = ops.layernorm(x, gamma, beta, 1e-5, bf16_copy=True, f32_out=False)
    assert torch.equal(only, o16)


@pytest.mark.parametrize("form", [0, 22, 21, 12, 11])
def test_layernorm_folded_around_the_gemm(dev, form):
    """pn_linear_ln_f32 (r6): a producer GEMM leaves (sum, sum of squares) per row and 32-column group of what it stores, the consumer GEMM
    applies LayerNorm to its input rows inside its own epilogue (set_transformer.py:160-165 `x + mlp(norm2(x))`).  Against float64:
    the statistics table to 1e-6 of its range, the consumer's output to 4e-6 of its range (tokens with a mean of the size of their spread,
    as residual streams have), every tile form, ragged rows, with GELU and residual."""
    from partner_amd import hip, ops
    for ci, (m, k, n, k0) in enumerate([(300, 256, 1024, 128), (1000, 256, 768, 256), (517, 128, 256, 64), (9000, 256, 512, 256)]):
        g = torch.Generator().manual_seed(100 * form + ci)
        x0 = torch.randn((m, k0), generator=g)
        w0 = torch.randn((k, k0), generator=g) / np.sqrt(k0)
        b0 = torch.randn((k,), generator=g) + 0.7          # a common offset: row means comparable to the spread
        res = torch.randn((m, k), generator=g)
        w1 = torch.randn((n, k), generator=g) / np.sqrt(k)
        b1 = torch.randn((n,), generator=g)
        res1 = torch.randn((m, n), generator=g)
        norm = torch.nn.LayerNorm(k, eps=1e-5)
        with torch.no_grad():
            norm.weight.copy_(1.0 + 0.3 * torch.randn((k,), generator=g))
            norm.bias.copy_(0.2 * torch.randn((k,), generator=g))
        prod = ops.GemmLayer(w0.to(dev), b0.to(dev))
        cons = ops.GemmLayer(w1.to(dev), b1.to(dev))
        assert prod.stats_ok and cons.fold_layernorm(norm.to(dev))
        hip.call("pn_linear_set_tile", form)
        try:
            y, st = prod(x0.to(dev), residual=res.to(dev), stats_out=True)
            z = cons(y, act=ops.ACT_GELU, residual=res1.to(dev), ln_stats=st)
            y_plain = prod(x0.to(dev), residual=res.to(dev))
            z_unfolded = cons(ops.layernorm(y_plain, norm.weight, norm.bias, norm.eps), act=ops.ACT_GELU, residual=res1.to(dev))
        finally:
            hip.call("pn_linear_set_tile", 0)
        assert torch.equal(y, y_plain)                      # the statistics epilogue does not touch the output
        yd = reference(x0, w0, b0, "none", res)
        ref_st = torch.stack([yd.view(m, k // 32, 32).sum(2), (yd * yd).view(m, k // 32, 32).sum(2)], 2)
        assert float((st.double().cpu() - ref_st).abs().max() / ref_st.abs().max()) < 1e-6
        zn = torch.nn.functional.layer_norm(yd, (k,), norm.weight.detach().double().cpu(), norm.bias.detach().double().cpu(), norm.eps)
        ref = reference(zn, w1, b1, "gelu", res1)
        err = float((z.double().cpu() - ref).abs().max() / ref.abs().max())
        err_unfolded = float((z_unfolded.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 4e-6 and err_unfolded < 4e-6, (form, m, k, n, err, err_unfolded)
