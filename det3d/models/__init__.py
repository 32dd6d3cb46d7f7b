from partner_amd.builder import (BACKBONES, BBOX_HEADS, DETECTORS, LOSSES, NECKS, READERS, ROI_HEAD, SECOND_STAGE,  # noqa: F401
                                 SEG_HEAD, build_backbone, build_bbox_head, build_detector, build_loss, build_neck,
                                 build_reader, build_roi_head, build_seg_head)
import partner_amd  # noqa: F401  (registers the modules)
