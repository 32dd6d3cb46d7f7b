"""partner_amd: MI355X-native (gfx950) implementation of PARTNER's polar-voxel encode -> BEV
backbone -> centre head hot path behind the det3d registry API.  All arithmetic runs in
hand-written HIP kernels (libpartner_hip.so); there is no CPU / PyTorch fallback."""
from .builder import (BACKBONES, BBOX_HEADS, DETECTORS, LOSSES, NECKS, READERS, ROI_HEAD, SECOND_STAGE, SEG_HEAD,  # noqa: F401
                      build_backbone, build_bbox_head, build_detector, build_loss, build_neck, build_reader, build_seg_head)
from .config import Config, ConfigDict, get_downsample_factor  # noqa: F401
from .registry import Registry, build_from_cfg  # noqa: F401
from . import readers, necks, necks_context, heads, swv_head, sparse_backbone, seg_heads, detectors  # noqa: F401,E402  (registers the modules)

__version__ = "0.1.0"
