"""Checkpoint compatibility: the `.tar` schema written by the reference's Logger.log
(src/utils/Logger.py:20-40) and consumed by get_mesh_tsdf_fusion.load_neural_point_cloud
(src/tools/get_mesh_tsdf_fusion.py:64-82) and eval_ate.py, with the positions coming from / going to the
device-resident index instead of a Python list of lists."""
from __future__ import annotations

import torch


def checkpoint_dict(npc, decoders, idx=0, keyframe_dict=None, keyframe_list=None, selected_keyframes=None,
                    gt_c2w_list=None, estimate_c2w_list=None, exposure_feat=None, positions_as_list=True) -> dict:
    """Same keys as Logger.log.  `cloud_pos` is a list of [x,y,z] lists as in the reference when
    positions_as_list (what the reference tools expect), else a [N,3] CPU tensor (cheaper at 1 M points)."""
    pos = npc.cloud_pos().detach().cpu()
    return {
        "geo_feats": npc.get_geo_feats().clone(),        # views of the pre-allocated stores: save N rows, not capacity
        "col_feats": npc.get_col_feats().clone(),
        "cloud_pos": pos.tolist() if positions_as_list else pos,
        "pts_num": npc.pts_num(),
        "input_pos": npc.input_pos(),
        "input_rgb": npc.input_rgb(),
        "decoder_state_dict": decoders.state_dict(),
        "gt_c2w_list": gt_c2w_list,
        "estimate_c2w_list": estimate_c2w_list,
        "keyframe_list": keyframe_list if keyframe_list is not None else [],
        "keyframe_dict": keyframe_dict if keyframe_dict is not None else [],
        "selected_keyframes": selected_keyframes if selected_keyframes is not None else {},
        "idx": idx,
        "exposure_feat_all": torch.stack(exposure_feat, dim=0) if exposure_feat is not None else None,
    }


def save_checkpoint(path, *args, **kwargs):
    torch.save(checkpoint_dict(*args, **kwargs), path)


def load_neural_point_cloud(npc, ckpt: dict):
    """get_mesh_tsdf_fusion.load_neural_point_cloud: positions + features into the npc, index rebuilt."""
    pos = ckpt["cloud_pos"]
    pos = pos if torch.is_tensor(pos) else torch.tensor(pos, dtype=torch.float32)
    ip, ir = ckpt.get("input_pos", []), ckpt.get("input_rgb", [])
    npc._input_pos = [torch.tensor(ip, dtype=torch.float32, device=npc.device).reshape(-1, 3)] if len(ip) else []
    npc._input_rgb = [torch.tensor(ir, dtype=torch.float32, device=npc.device).reshape(-1, 3)] if len(ir) else []
    npc.geo_feats = None
    npc.col_feats = None
    npc.set_points(pos.reshape(-1, 3), ckpt["geo_feats"], ckpt["col_feats"])
    return npc.pts_num()


def load_decoders(decoders, ckpt: dict):
    """decoder_state_dict (reference POINT keys).  The fixed colour Fourier matrix is NOT part of any reference
    checkpoint (decoder.py:27-28,305-306): it is reproduced by constructing the decoders under the same seed."""
    missing, unexpected = decoders.load_state_dict(ckpt["decoder_state_dict"], strict=False)
    return missing, unexpected
