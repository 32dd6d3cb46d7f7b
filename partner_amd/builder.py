"""Registries and builders named as in the reference (det3d/models/registry.py:3-11,
det3d/models/builder.py:17-53)."""
from torch import nn

from .registry import Registry, build_from_cfg

READERS = Registry("reader")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
BBOX_HEADS = Registry("bbox_heads")
SEG_HEAD = Registry("seg_heads")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
SECOND_STAGE = Registry("second_stage")
ROI_HEAD = Registry("roi_head")


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_reader(cfg):
    return build(cfg, READERS)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_bbox_head(cfg):
    return build(cfg, BBOX_HEADS)


def build_seg_head(cfg):
    return build(cfg, SEG_HEAD)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_roi_head(cfg):
    return build(cfg, ROI_HEAD)


def build_second_stage_module(cfg):
    return build(cfg, SECOND_STAGE)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
