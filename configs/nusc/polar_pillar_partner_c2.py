# nuScenes polar-pillar model of BASELINE.json configs[1] in the det3d config schema (own file,
# written from the parameter digest in SURVEY.md Appendix A.1).  It is loaded exactly like a
# reference config:  Config.fromfile(...) -> build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
import itertools
import logging

from det3d.utils.config_tool import get_downsample_factor

voxel_generator = dict(
    range=[0.3, -3.1488, -5.0, 50.476, 3.1488, 3.0],
    voxel_size=[0.098, 0.0123, 8],
    max_points_in_voxel=20,
    max_voxel_num=[30000, 60000],
    voxel_shape="cylinder",
    return_density=True,
    dynamic=True,
    nsectors=1,
)
tasks = [dict(num_class=10, class_names=["car", "truck", "construction_vehicle", "bus", "trailer", "barrier",
                                         "motorcycle", "bicycle", "pedestrian", "traffic_cone"])]
class_names = list(itertools.chain(*[t["class_names"] for t in tasks]))

model = dict(
    type="PointPillars",
    pretrained=None,
    reader=dict(type="DynamicPFNet", num_filters=[64, 128], num_input_features=7, voxel_shape="cylinder",
                xyz_cluster=True, raz_cluster=True, xy_center=True, ra_center=True,
                voxel_size=voxel_generator["voxel_size"], pc_range=voxel_generator["range"]),
    backbone=dict(type="DynamicPPScatter", ds_factor=1),
    neck=dict(type="RPN", layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[128, 128, 256],
              us_layer_strides=[0.5, 1, 2], us_num_filters=[128, 128, 128], num_input_features=128,
              logger=logging.getLogger("RPN")),
    bbox_head=dict(type="CenterHeadSinglePos", in_channels=sum([128, 128, 128]), tasks=tasks, dataset="nuscenes",
                   weight=0.5, code_weights=[1.5, 1.5, 1.0, 1.0, 1.0, 1.0, 0.5, 0.5, 1.0, 1.0],
                   common_heads={"reg": (2, 2), "rot_vel": (2, 2), "height": (1, 2), "dim": (3, 2)},
                   voxel_shape=voxel_generator["voxel_shape"], voxel_generator=voxel_generator),
    seg_head=None,
    part_head=None,
)
train_cfg = dict(assigner=dict(out_size_factor=get_downsample_factor(model), gaussian_overlap=0.1, max_objs=500, min_radius=2))
# the reference config's post-processing settings (polarstream_det_n_seg_1_sector.py:102-116): per-class rotated NMS
rectify = True
test_cfg = dict(
    post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
    max_per_img=500,
    per_class_nms=True,
    rectify=rectify,
    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.1),
    score_threshold=0.1,
    pc_range=voxel_generator["range"],
    out_size_factor=get_downsample_factor(model),
    voxel_size=voxel_generator["voxel_size"],
)
