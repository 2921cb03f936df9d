"""Configuration surface: the reference's YAML files are consumed as they are.

`load_config` follows the reference's `inherit_from` chain (src/config.py:5-51);
`default_config()` is the hot-path subset of configs/point_slam.yaml restated as
a dict so that the package works without the reference tree (tests, bench)."""
from __future__ import annotations

import copy

import yaml


def update_recursive(dst: dict, src: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            update_recursive(dst[k], v)
        else:
            dst[k] = v


def load_config(path: str, default_path: str = None) -> dict:
    with open(path, "r") as f:
        special = yaml.full_load(f)
    parent = special.get("inherit_from")
    if parent is not None:
        cfg = load_config(parent, default_path)
    elif default_path is not None:
        with open(default_path, "r") as f:
            cfg = yaml.full_load(f)
    else:
        cfg = default_config()
    update_recursive(cfg, special)
    return cfg


_DEFAULT = {
    "use_dynamic_radius": True, "setup_seed": 1219,
    "model": {"c_dim": 32, "exposure_dim": 8, "pos_embedding_method": "fourier", "encode_rel_pos_in_col": True,
              "encode_exposure": False, "use_view_direction": False, "encode_viewd": True},
    "tracking": {"ignore_edge_W": 20, "ignore_edge_H": 20, "use_color_in_tracking": True, "device": "cuda:0",
                 "handle_dynamic": True, "depth_limit": False, "w_color_loss": 0.5, "separate_LR": True,
                 "const_speed_assumption": True, "sample_with_color_grad": False, "gt_camera": False, "lr": 0.002,
                 "pixels": 200, "iters": 20},
    "mapping": {"device": "cuda:0", "color_refine": True, "geo_iter_ratio": 0.4, "geo_iter_first": 400,
                "every_frame": 5, "BA": False, "BA_cam_lr": 0.0002, "frustum_edge": -4, "fix_geo_decoder": True,
                "fix_color_decoder": False, "keyframe_every": 50, "mapping_window_size": 5, "w_color_loss": 0.1,
                "frustum_feature_selection": True, "keyframe_selection_method": "overlap", "pixels": 1000,
                "pixels_adding": 6000, "pixels_based_on_color_grad": 0, "iters_first": 1500, "iters": 400,
                "min_iter_ratio": 0.95,
                "init": {"geometry": {"decoders_lr": 0.001, "geometry_lr": 0.03, "color_lr": 0.0},
                         "color": {"decoders_lr": 0.005, "geometry_lr": 0.005, "color_lr": 0.005}},
                "stage": {"geometry": {"decoders_lr": 0.001, "geometry_lr": 0.03, "color_lr": 0.0},
                          "color": {"decoders_lr": 0.005, "geometry_lr": 0.005, "color_lr": 0.005}}},
    "cam": {"H": 480, "W": 640, "fx": 517.0, "fy": 517.0, "cx": 319.5, "cy": 239.5, "crop_edge": 0},
    "rendering": {"N_surface": 5, "near_end": 0.3, "near_end_surface": 0.98, "far_end_surface": 1.02,
                  "sigmoid_coef_tracker": 0.1, "sigmoid_coef_mapper": 0.1, "sample_near_pcl": True},
    "pointcloud": {"nn_num": 8, "min_nn_num": 2, "N_add": 3, "nn_weighting": "distance", "radius_add": 0.04,
                   "radius_min": 0.02, "radius_query": 0.08, "radius_add_max": 0.08, "radius_add_min": 0.02,
                   "radius_query_ratio": 2, "color_grad_threshold": 0.15, "near_end_surface": 0.98,
                   "far_end_surface": 1.02, "nlist": 400, "nprobe": 4, "fix_interval_when_add_along_ray": False},
}


def default_config() -> dict:
    return copy.deepcopy(_DEFAULT)


def replica_overrides(cfg: dict) -> dict:
    """configs/Replica/replica.yaml:7-17 iteration mix."""
    cfg = copy.deepcopy(cfg)
    cfg["tracking"].update(pixels=1500, iters=40, ignore_edge_W=100, ignore_edge_H=100)
    cfg["mapping"].update(pixels=5000, iters=300, mapping_window_size=12, keyframe_every=20,
                          pixels_based_on_color_grad=1000)
    return cfg


def tum_overrides(cfg: dict) -> dict:
    """configs/TUM_RGBD/tum.yaml: noisy-depth path, 5 000 tracking / 10 000 mapping pixels per iteration, pixel sets
    drawn from the top colour gradients, plain feature interpolation (no per-neighbour colour MLP)."""
    cfg = copy.deepcopy(cfg)
    cfg["model"].update(encode_rel_pos_in_col=False)
    cfg["tracking"].update(separate_LR=False, pixels=5000, iters=200, sample_with_color_grad=True)
    cfg["mapping"].update(every_frame=2, mapping_window_size=10, pixels=10000, iters_first=500, geo_iter_first=200,
                          iters=150)
    return cfg


def scannet_overrides(cfg: dict) -> dict:
    """configs/ScanNet/scannet.yaml: per-frame exposure latents, wider sampling interval, window of 20."""
    cfg = copy.deepcopy(cfg)
    cfg["model"].update(encode_exposure=True, encode_rel_pos_in_col=False, encode_viewd=False)
    cfg["tracking"].update(separate_LR=False, lr=0.0005, pixels=5000, iters=100, sample_with_color_grad=True)
    cfg["mapping"].update(geo_iter_ratio=0.3, mapping_window_size=20, keyframe_every=10, pixels=10000, iters_first=500,
                          geo_iter_first=200, iters=300)
    cfg["rendering"].update(near_end_surface=0.96, far_end_surface=1.04)
    cfg["pointcloud"].update(near_end_surface=0.96, far_end_surface=1.04)
    cfg["cam"].update(crop_edge=10)
    return cfg


MIXES = {"base": lambda c: copy.deepcopy(c), "replica": replica_overrides, "tum": tum_overrides,
         "scannet": scannet_overrides}
