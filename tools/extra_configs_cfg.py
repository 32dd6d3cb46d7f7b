"""head section of configs/waymo/voxelnet/waymo_partner_36epoch.py as a plain dict (shared by tests and tools)"""


def waymo_head_cfg():
    tasks = [dict(num_class=1, class_names=["Vehicle"])]
    vg = dict(range=[0.3, -3.14368, -2.0, 75.18, 3.14368, 4.0], voxel_size=[0.065, 0.00307, 0.15], max_points_in_voxel=5, max_voxel_num=150000,
              voxel_shape="cylinder", return_density=False, dynamic=False, nsectors=1)
    return dict(
        type="E2ESWVoteHead", in_channels=512, tasks=tasks, dataset="waymo", weight=2, code_weights=[1.0] * 8,
        common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)}, voxel_shape="cylinder", voxel_generator=vg, out_size_factor=8,
        SET_CRIT_CONFIG={"weight_dict": {"loss_ce": 1, "loss_bbox": 2, "loss_vote": 0.25, "loss_vote_cls": 1, "loss_iou": 2},
                         "losses": ["loss_ce", "loss_bbox", "loss_vote", "loss_vote_cls", "loss_iou"], "sigma": 3.0, "code_weights": [1.0] * 8,
                         "use_focal_loss": True, "gamma": 2.0, "alpha": 0.25},
        CODER_CONFIG={"code_size": 7, "encode_angle_by_sincos": True},
        MATCHER_CONFIG={"weight_dict": {"loss_ce": 0.25, "loss_bbox": 0.75}, "losses": ["loss_ce", "loss_bbox"], "code_weights": [1.0] * 8,
                        "use_focal_loss": True, "box_pred_metric": "loss_bbox", "use_heatmap": False},
        USE_FOCAL_LOSS=True,
        GT_PROCESSOR_CONFIG={"tasks": tasks, "generate_votemap": True, "max_volumn_space": [75.18, 3.14368, 4.0],
                             "min_volumn_space": [0.3, -3.14368, -2.0], "grid_size": [1152, 2048, 40], "feature_map_stride": 8, "gaussian_overlap": 0.1,
                             "min_radius": 4, "num_max_objs": 500, "scale_factor": 2, "mapping": {"Vehicle": 1}},
        HEAD_CONFIG={"kernel_size": 3, "sw_head_version": "votev4", "cls_head_version": "v2", "window_size": 7, "sl_depth": [2], "code_size": 7,
                     "encode_angle_by_sincos": True, "iou_loss": True, "iou_factor": 1, "init_bias": -2.19, "num_classes": 1})
