"""Tensor-level wrappers over the C ABI (device memory from PyTorch, kernels from libpartner_hip).

Activations are NHWC fp32 tensors of shape (B, H, W, C); ``as_nchw`` / ``to_nhwc`` convert at the
det3d API boundary (a channels-last NCHW view is free).  Nothing here touches the oracle and
nothing falls back to PyTorch arithmetic.

r6: the wrappers live in stage modules -- ``ops_index`` (frame index, scatter stage, voxelization, sectors), ``ops_conv`` (convolutions),
``ops_token`` (token GEMMs, LayerNorm), ``ops_train`` (backward operators, loss, optimizer kernels), ``ops_common`` (layout, scratch,
streams) -- and this module re-exports them all, so ``ops.X`` keeps working everywhere.  Route switches: ``ops.R`` (routes.Routes, ONE
object read at call time by every stage module: ``with ops.R.override(linear=False): ...``); launch state (frames-in-flight hint, chain
form of the frame being captured, the conv profiler): ``ops.S``.
"""
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, ConvDesc  # noqa: F401
from .ops_common import *  # noqa: F401,F403
from .ops_common import _f32, _workspace  # noqa: F401
from .ops_conv import *  # noqa: F401,F403
from .ops_conv import _chain_orientation  # noqa: F401
from .ops_index import *  # noqa: F401,F403
from .ops_token import *  # noqa: F401,F403
from .ops_train import *  # noqa: F401,F403
from .routes import R, S  # noqa: F401
