"""Every switch of the Python layer in ONE object, and the process-wide launch state in another (r6).

``R`` (``Routes``): the kernel-route switches, read once from the ``PN_*`` environment variables at import (README.md, "environment
switches", lists them; ``tests/test_hip_routes.py`` runs the parity subset under four alternate sets).  Up to r5 these were module-level
globals scattered over ``ops.py``, ``heads.py``, ``sparse_backbone.py`` and ``train.py``; a test that wanted another route inside one
interpreter rebound the global of the module that read it, which is why ``ops.py`` could not be split.  Now every module reads
``routes.R.<name>`` at CALL time, so ``R.override(linear=False)`` (a context manager) or a plain attribute assignment switches the route
for every stage module at once.

``S`` (``State``): what a launch consults besides its arguments -- the frames-in-flight hint, the chain-form route of the frame being
captured, the conv profiler of bench.py's roofline pass.  Engines are captured from one thread; nothing here is thread-local.
"""
from __future__ import annotations

import os
from contextlib import contextmanager


class Routes:
    def __init__(self):
        # ---- convolutions (ops_conv.py)
        self.conv_wino = os.environ.get("PN_CONV_WINO", "1") != "0"                  # width-Winograd F(2, 3) for 3x3 / stride-1 layers
        self.conv_wino_min_tiles = int(os.environ.get("PN_CONV_WINO_MIN_TILES", "256"))
        # F(4, 3) (conv_wino4.hip): 0 keeps F(2, 3); taken from this many 32-quad x 32-column tiles on
        self.conv_wino4 = os.environ.get("PN_CONV_WINO4", "1") != "0"
        self.conv_wino4_min_tiles = int(os.environ.get("PN_CONV_WINO4_MIN_TILES", "256"))
        self.conv_wino4_dgrad = os.environ.get("PN_CONV_WINO4_DGRAD", "1") != "0"    # F(4, 3) for the training data gradients
        self.conv_tapsum = os.environ.get("PN_CONV_TAPSUM", "1") != "0"              # 3x3 layers with <= 3 outputs over >= 128 inputs as GEMM + tap sum
        self.conv_chain = os.environ.get("PN_CONV_CHAIN", "1") != "0"                # same-map 3x3 / stride-1 runs kept in the F(4, 3) domain (conv_wchain.hip)
        self.conv_chain2d = os.environ.get("PN_CONV_CHAIN2D", "1") != "0"            # ... with F(2, 3) along the height on top
        self.conv_chain44 = os.environ.get("PN_CONV_CHAIN44", "1") != "0"            # ... with F(4, 3) along the height where other frames are in flight
        self.conv_planes = os.environ.get("PN_CONV_PLANES", "1") != "0"              # a block's stride-2 layer writes the chain's planes itself (r6)
        self.conv_wgrad_wino4 = os.environ.get("PN_CONV_WGRAD_WINO4", "1") != "0"    # F(4, 3) weight gradients
        self.conv_wgrad_wino4_min_quads = int(os.environ.get("PN_CONV_WGRAD_WINO4_MIN_QUADS", "4096"))
        # ---- first convolution on the pillar canvas
        self.pillar_conv = os.environ.get("PN_PILLAR_CONV", "1") != "0"              # (pillar, tap) forms at all
        self.pillar_planes = os.environ.get("PN_PILLAR_PLANES", "1") != "0"          # ... writing the chain's planes directly
        self.pillar_rows = os.environ.get("PN_PILLAR_ROWS", "1") != "0"              # ... as the row-band kernel (pillar_rows.hip, r6); 0: pair lists
        self.pillar_conv_max_fill = float(os.environ.get("PN_PILLAR_CONV_MAX_FILL", "0.35"))
        self.pillar_rows_max_fill = float(os.environ.get("PN_PILLAR_ROWS_MAX_FILL", "1.25"))
        self.fused_sweeps = os.environ.get("PN_FUSED_SWEEPS", "1") != "0"            # streaming frames: sweep accumulation inside the frame index (r6)
        self.sample_streams = os.environ.get("PN_SAMPLE_STREAMS", "1") != "0"        # VoxelNetV3: the dense stages of a batch per sample on two streams (r6)
        self.pfn_clears_index = os.environ.get("PN_PFN_CLEARS_INDEX", "1") != "0"    # frame engines: the reader's launch zeroes the frame's index counters (r6)
        # ---- token GEMMs
        self.linear = os.environ.get("PN_LINEAR", "1") != "0"                        # 0: the r2 route (1x1 convolution on conv_mfma_kernel)
        self.ln_fold = os.environ.get("PN_LN_FOLD", "1") != "0"                      # LayerNorm folded into the consuming GEMM (r6)
        # ---- heads / sparse encoder
        self.head_chain = os.environ.get("PN_HEAD_CHAIN", "1") != "0"                # first-stage branch convolutions chained in the Winograd domain
        self.sparse_c16 = os.environ.get("PN_SPARSE_C16", "1") != "0"                # 0: the 16-channel level on the gathered MFMA kernel as well
        self.sparse_row_bits = os.environ.get("PN_SPARSE_ROW_BITS", "1") != "0"      # the grouping sort reads per-row tap bytes left by the neighbour kernel
        self.sparse_grouped = os.environ.get("PN_SPARSE_GROUPED", "1") != "0"        # 0: every level >= 32 channels on the gathered tile kernel (r3)
        self.sparse_struct_stream = os.environ.get("PN_SPARSE_STRUCT_STREAM", "1") != "0"   # 0: index builds / neighbour tables on the calling stream
        # ---- training
        self.train_wgrad_stream = os.environ.get("PN_TRAIN_WGRAD_STREAM", "1") != "0"      # weight gradients on a second stream
        self.train_wino4_max_pixels = int(os.environ.get("PN_TRAIN_WINO4_MAX_PIXELS", "16384"))
        self.train_pillar_conv = os.environ.get("PN_TRAIN_PILLAR_CONV", "1") != "0"
        self.train_prepack = os.environ.get("PN_TRAIN_PREPACK", "1") != "0"
        self.train_chain_max_pixels = int(os.environ.get("PN_TRAIN_CHAIN_MAX_PIXELS", "4096"))
        self.train_strat_expand = os.environ.get("PN_TRAIN_STRAT_EXPAND", "0") != "0"

    @contextmanager
    def override(self, **kw):
        """``with R.override(linear=False, conv_chain=False): ...`` -- routes for the block, restored on exit (also when it raises)"""
        old = {k: getattr(self, k) for k in kw}      # (AttributeError for a name that is not a switch)
        try:
            for k, v in kw.items():
                setattr(self, k, v)
            yield self
        finally:
            for k, v in old.items():
                setattr(self, k, v)


class State:
    def __init__(self):
        self.frames_in_flight = 1      # hint of the frame being launched / captured (ops.frames_in_flight)
        self.chain44_route = True      # False: the frame takes F(2,3)xF(4,3) where the hint alone would pick F(4,3)xF(4,3) (ops.chain44)
        self.chain44_launches = 0      # launches that took the F(4,3)xF(4,3) form so far (engine.FramePipeline: is there a choice to measure?)
        self.profiler = None           # ops.ConvProfiler while bench.py's roofline pass runs
        # weight layouts / plans built lazily so far (hip.call("pn_pack_*"), nn_utils.PlanCache rebuilds): each is queued on the stream that first
        # needs it -- a caller that spreads work over several streams looks at this counter to see whether anything was built during its
        # first stream's launches and, if so, lets the other streams wait for it (VoxelNetV3.dense_stages_nhwc)
        self.lazy_builds = 0


R = Routes()
S = State()
