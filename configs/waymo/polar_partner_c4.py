# Waymo polar PARTNER model of BASELINE.json configs[3] in the det3d config schema (own file, written from the
# parameter digest in SURVEY.md Appendix A.2; the reference's configs/waymo/voxelnet/waymo_partner_36epoch.py
# loads through the same Config / build_detector path unchanged).  Model section and the post-processing
# thresholds only: dataset / pipeline / schedule sections are outside the hot path.
import logging

from det3d.utils.config_tool import get_downsample_factor

polar_range = [0.3, -3.14368, -2.0, 75.18, 3.14368, 4.0]        # rho, azimuth, z
polar_voxel = [0.065, 0.00307, 0.15]                             # -> 1152 x 2048 x 40 cells
cells = [1152, 2048, 40]
voxel_generator = dict(range=polar_range, voxel_size=polar_voxel, max_points_in_voxel=5, max_voxel_num=150000, voxel_shape="cylinder",
                       return_density=False, dynamic=False, nsectors=1)
tasks = [dict(num_class=1, class_names=["Vehicle"])]

swin_head = dict(kernel_size=3, sw_head_version="votev4", cls_head_version="v2", window_size=7, sl_depth=[2], code_size=7,
                 encode_angle_by_sincos=True, iou_loss=True, iou_factor=1, init_bias=-2.19, num_classes=1)
targets = dict(tasks=tasks, generate_votemap=True, max_volumn_space=polar_range[3:], min_volumn_space=polar_range[:3], grid_size=cells,
               feature_map_stride=8, gaussian_overlap=0.1, min_radius=4, num_max_objs=500, scale_factor=2, mapping={"Vehicle": 1})

model = dict(
    type="VoxelNetV3",
    pretrained=None,
    reader=dict(type="VoxelFeatureExtractorV3", num_input_features=7),
    backbone=dict(type="SpMiddleResNetFHD", num_input_features=7, ds_factor=8),
    neck=dict(type="RPN", layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256], us_layer_strides=[1, 2],
              us_num_filters=[256, 256], num_input_features=256, logger=logging.getLogger("RPN")),
    bbox_head=dict(type="E2ESWVoteHead", in_channels=256 + 256, tasks=tasks, dataset="waymo", weight=2, code_weights=[1.0] * 8,
                   common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)}, voxel_shape="cylinder",
                   voxel_generator=voxel_generator, out_size_factor=8, USE_FOCAL_LOSS=True,
                   CODER_CONFIG=dict(code_size=7, encode_angle_by_sincos=True), GT_PROCESSOR_CONFIG=targets, HEAD_CONFIG=swin_head),
    seg_head=None,
    part_head=None,
)
train_cfg = dict(assigner=dict(out_size_factor=get_downsample_factor(model), gaussian_overlap=0.1, max_objs=500, min_radius=2))
# post-processing settings of the reference config (waymo_partner_36epoch.py:142-154): multi-class rotate_nms_pcdet
rectify = False
test_cfg = dict(
    post_center_limit_range=[-80, -80, -10.0, 80, 80, 10.0],
    nms=dict(nms_pre_max_size=4096, nms_post_max_size=500, nms_iou_threshold=0.7),
    score_threshold=0.1,
    pc_range=voxel_generator["range"],
    out_size_factor=get_downsample_factor(model),
    voxel_size=voxel_generator["voxel_size"],
    rectify=rectify,
)
