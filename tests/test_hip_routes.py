"""The alternate kernel routes behind the product's PN_* environment switches (README "environment switches"): each set below switches a
family of r2 - r4 routes back on -- the switches are read at import time, so every set runs in a fresh interpreter -- and the SAME parity
tests the default routes pass (oracle / reference goldens, full-size C2 model included) must pass on them.  A route that no longer agrees
with the oracle fails here instead of waiting for someone to flip its switch (only default combinations were tested)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the parity subset: reduced model stage by stage + the full-size C2 model against the reference's tensors, both heads, the SetBlock against the
# reference golden, the sparse encoder and the geometry-aware head against the oracle, one training step against the reference's gradients
SUBSET = ["tests/test_hip_model.py::test_small_model_stage_by_stage", "tests/test_hip_model.py::test_full_c2_model",
          "tests/test_hip_model.py::test_center_head_plain_and_single", "tests/test_hip_model.py::test_fused_head_vs_oracle_and_unfused",
          "tests/test_hip_attention.py::test_setblock_small", "tests/test_hip_sparse.py::test_sp_middle_resnet_fhd_matches_oracle",
          "tests/test_hip_swv.py::test_e2e_swv_head_matches_oracle", "tests/test_hip_train.py::test_small_model_train_step_grads",
          "tests/test_hip_train.py::test_small_model_train_steps_run",
          # r6: the frame engines (the reader's launch clearing the index counters, sweeps inside the index launch) and the per-sample streams
          "tests/test_hip_model.py::test_frame_engine_keeps_a_dirty_canvas_only_where_nothing_reads_it",
          "tests/test_hip_model.py::test_streaming_engine_raw_sweeps_to_boxes", "tests/test_hip_sparse.py::test_voxelnet_v3_batch_of_two"]

ROUTES = {
    # r3 and earlier forms of everything r4 / r5 replaced: one F(4,3) launch per layer instead of the chains, the head's multi-job launches,
    # token GEMMs as 1x1 convolutions, gathered 128-site sparse tiles, the dense first layer, the direct small-N head convolutions
    "r3_forms": dict(PN_CONV_CHAIN="0", PN_HEAD_CHAIN="0", PN_LINEAR="0", PN_SPARSE_GROUPED="0", PN_SPARSE_C16="0", PN_PILLAR_CONV="0",
                     PN_CONV_TAPSUM="0", PN_SMALL_N_TWO="0", PN_TRAIN_PILLAR_CONV="0", PN_TRAIN_PREPACK="0", PN_PFN_BWD_SINGLE="0"),
    # no Winograd anywhere: every convolution on the direct implicit-GEMM kernel, weight gradients on the direct kernel, one training stream
    "direct_only": dict(PN_CONV_WINO="0", PN_CONV_WINO4="0", PN_CONV_CHAIN="0", PN_HEAD_CHAIN="0", PN_CONV_WGRAD_WINO4="0", PN_CONV_WINO4_DGRAD="0",
                        PN_CONV_SMALL_N="0", PN_TRAIN_WGRAD_STREAM="0", PN_TRAIN_CHAIN_MAX_PIXELS="0"),
    # the one-dimensional chain, F(2,3) without F(4,3), the wave-per-group sparse kernel, row bits off, planes off
    "r4_alternates": dict(PN_CONV_CHAIN2D="0", PN_SPARSE_GROUP4="0", PN_SPARSE_ROW_BITS="0", PN_PILLAR_PLANES="0", PN_WINO4_KSPLIT="0",
                          PN_TRAIN_STRAT_EXPAND="1", PN_SPARSE_G4SPLIT="0", PN_PFN_SPLIT="0", PN_CHANNEL_SUM_V4="0", PN_SMALL_N_MFMA="0", PN_CONV_CHAIN44="0", PN_PFN_TILES="0",
                          PN_PILLAR_ROWS="0", PN_LN_FOLD="0",       # (r6: the pair-list first convolution, LayerNorm passes in front of the token GEMMs,
                          PN_SAMPLE_STREAMS="0", PN_PFN_CLEARS_INDEX="0", PN_FUSED_SWEEPS="0", PN_CONV_PLANES="0"),      # one launch sequence per batch, a clear launch per frame, ...)
    "f23_only": dict(PN_CONV_WINO4="0", PN_CONV_CHAIN="0", PN_HEAD_CHAIN="0", PN_WINO_BDIRECT="0", PN_SPARSE_WINDOW="1024", PN_WINO4_TWO_PHASE="0"),
}


@pytest.mark.parametrize("name", sorted(ROUTES))
def test_alternate_routes_pass_the_parity_subset(name):
    env = dict(os.environ, **ROUTES[name])
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + SUBSET, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, f"{name}: {ROUTES[name]}\n" + r.stdout[-3000:] + r.stderr[-1000:]
