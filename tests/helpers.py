"""Shared test helpers: fixture loading + the reference config variants used by the fixtures."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def base_cfg():
    """The hot-path subset of configs/point_slam.yaml (reference), restated as a dict."""
    return {
        "use_dynamic_radius": True,
        "setup_seed": 1219,
        "model": {"c_dim": 32, "exposure_dim": 8, "pos_embedding_method": "fourier",
                  "encode_rel_pos_in_col": True, "encode_exposure": False, "use_view_direction": False,
                  "encode_viewd": True},
        "tracking": {"ignore_edge_W": 20, "ignore_edge_H": 20, "use_color_in_tracking": True,
                     "handle_dynamic": True, "w_color_loss": 0.5, "separate_LR": True, "lr": 0.002,
                     "pixels": 200, "iters": 20, "device": "cuda:0", "depth_limit": False,
                     "sample_with_color_grad": False},
        "mapping": {"w_color_loss": 0.1, "pixels": 1000, "iters": 400, "every_frame": 5, "device": "cuda:0",
                    "geo_iter_ratio": 0.4, "fix_geo_decoder": True, "fix_color_decoder": False,
                    "mapping_window_size": 5, "pixels_adding": 6000, "frustum_edge": -4, "BA": False,
                    "min_iter_ratio": 0.95, "geo_iter_first": 400, "iters_first": 1500, "keyframe_every": 20,
                    "pixels_based_on_color_grad": 0, "keyframe_selection_method": "overlap",
                    "init": {"geometry": {"decoders_lr": 0.001, "geometry_lr": 0.03, "color_lr": 0.0},
                             "color": {"decoders_lr": 0.005, "geometry_lr": 0.005, "color_lr": 0.005}},
                    "stage": {"geometry": {"decoders_lr": 0.001, "geometry_lr": 0.03, "color_lr": 0.0},
                              "color": {"decoders_lr": 0.005, "geometry_lr": 0.005, "color_lr": 0.005}}},
        "rendering": {"N_surface": 5, "near_end": 0.3, "near_end_surface": 0.98, "far_end_surface": 1.02,
                      "sigmoid_coef_tracker": 0.1, "sigmoid_coef_mapper": 0.1, "sample_near_pcl": True},
        "pointcloud": {"nn_num": 8, "min_nn_num": 2, "N_add": 3, "nn_weighting": "distance",
                       "radius_add": 0.04, "radius_min": 0.02, "radius_query": 0.08, "radius_add_max": 0.08,
                       "radius_add_min": 0.02, "radius_query_ratio": 2, "color_grad_threshold": 0.15,
                       "near_end_surface": 0.98, "far_end_surface": 1.02, "nlist": 400, "nprobe": 4,
                       "fix_interval_when_add_along_ray": False},
        "cam": {"crop_edge": 0},
    }


def cfg_variant(name):
    cfg = base_cfg()
    if name == "tum":
        cfg["use_dynamic_radius"] = False
        cfg["model"]["encode_rel_pos_in_col"] = False
    elif name == "replica_expo":     # pointcloud.nn_weighting = 'expo' (decoder.py:154-156, 364-366; no shipped config)
        cfg["pointcloud"]["nn_weighting"] = "expo"
    elif name == "scannet":
        cfg["model"]["encode_rel_pos_in_col"] = False
        cfg["model"]["encode_exposure"] = True
        cfg["rendering"]["near_end_surface"] = 0.96
        cfg["rendering"]["far_end_surface"] = 1.04
    return cfg


def load_npz(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        if a.dtype.kind in "US":
            out[k] = str(a)
        elif a.shape == () and a.dtype.kind in "b":
            out[k] = bool(a)
        elif a.shape == () and a.dtype.kind in "fi":
            out[k] = a.item()
        else:
            out[k] = torch.from_numpy(a)
    return out


def load_decoders(cfg_name):
    which = "scannet" if cfg_name == "scannet" else "replica"
    return load_npz("decoders_seed1219_" + which)


def fixture_cfg(fx):
    """Config of a render fixture: the named variant + the flags the generator recorded."""
    cfg = cfg_variant(fx["cfg_name"])
    if "sample_near_pcl" in fx:
        cfg["rendering"]["sample_near_pcl"] = bool(fx["sample_near_pcl"])
    return cfg


# the last two hold pixels without sensor depth (sample_near_pcl / uniform branch, Renderer.py:142-170)
# render_scannet_color_mapper: the ScanNet mapper call -- encode_exposure with exposure_feat=None returns the raw colour
# logits (PSL_NO_SIGMOID); the per-frame affine + sigmoid are applied by the caller (decoder.py:432-448, Mapper.py:530-548)
RENDER_CASES = ["render_replica_color_tracker", "render_replica_color_mapper", "render_replica_geometry_mapper",
                "render_tum_color_mapper", "render_scannet_color_tracker", "render_holes_nearpcl_mapper",
                "render_holes_uniform_tracker", "render_scannet_color_mapper",
                # nn_weighting = 'expo' (round 6): the mapper's two stages; the reference's TRACKER raises with it (gen_golden.py)
                "render_expo_color_mapper", "render_expo_geometry_mapper"]
ORACLE_ONLY_CASES = []


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


# ---- loop fixtures (oracle/gen_golden_loops.py: the reference's own Mapper.optimize_map / Tracker loops) -------
def loop_cfg(fx):
    """Config of a mapper_iters_* / tracker_iters_* fixture."""
    cfg = cfg_variant(fx["cfg_name"])
    if "mapping_pixels" in fx:
        cfg["mapping"].update(pixels=fx["mapping_pixels"], iters=fx["iters_cfg"], pixels_adding=fx["pixels_adding"],
                              mapping_window_size=3)
    return cfg


def loop_cam(fx):
    from point_slam_amd import synthetic as syn
    return syn.intrinsics(fx["W"], fx["H"])


def mapper_frames(fx, exposure):
    return [dict(depth=fx[f"f{k}_depth"], color=fx[f"f{k}_color"], c2w=fx[f"f{k}_c2w"], r_query=fx[f"f{k}_r_query"],
                 exposure=fx[f"f{k}_exposure"] if exposure else None) for k in range(3)]
