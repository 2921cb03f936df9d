"""Parameter container with the reference decoder's state_dict surface.

The reference POINT module (src/conv_onet/models/decoder.py:452-518) both OWNS
the weights and EVALUATES them with ATen ops.  Here evaluation happens in the
HIP kernels, so this module only owns parameters under the same state_dict keys
(checkpoints written by src/utils/Logger.py:22-40 load with strict=False) and
the same initialisation rules (xavier-uniform with relu gain for the trunk,
decoder.py:40-52; default nn.Linear init for fc_c; N(0, scale^2) Fourier
matrices, decoder.py:24-28).  Unused reference tensors (geo_decoder's
mlp_col_neighbor / embedder_rel_pos, decoder.py:108-111) are kept so key sets match.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class _Fourier(nn.Module):
    def __init__(self, mapping_size, scale, learnable):
        super().__init__()
        B = torch.randn((3, mapping_size)) * scale
        if learnable:
            self._B = nn.Parameter(B)
        else:
            self._B = B           # plain attribute: NOT in state_dict, exactly like the reference


class _Dense(nn.Linear):
    def __init__(self, i, o, activation="relu"):
        self.activation = activation
        super().__init__(i, o)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight, gain=nn.init.calculate_gain(self.activation))
        nn.init.zeros_(self.bias)


class _ColNeighbor(nn.Module):
    def __init__(self, c_dim, emb, hidden):
        super().__init__()
        self.linear1 = nn.Linear(c_dim + emb, hidden)
        self.linear2 = nn.Linear(hidden, c_dim)
        nn.init.xavier_uniform_(self.linear1.weight)
        nn.init.xavier_uniform_(self.linear2.weight)


class _Exposure(nn.Module):
    """MLP_exposure (decoder.py:243-258): 8 -> 128 -> 12; evaluated in torch (one vector per batch)."""

    def __init__(self, latent, hidden):
        super().__init__()
        self.linear1 = nn.Linear(latent, hidden)
        self.linear2 = nn.Linear(hidden, 12)
        self.act_fn = nn.Softplus(beta=100)
        nn.init.normal_(self.linear1.weight, mean=0, std=0.01)
        nn.init.normal_(self.linear2.weight, mean=0, std=0.01)

    def forward(self, x):
        return self.linear2(self.act_fn(self.linear1(x)))


class _Decoder(nn.Module):
    def __init__(self, cfg, hidden, emb_size, emb_scale, emb_concat, emb_learnable, out_dim, out_act, exposure):
        super().__init__()
        c_dim = cfg['model']['c_dim']
        self.fc_c = nn.ModuleList([nn.Linear(c_dim, hidden) for _ in range(5)])
        self.embedder = _Fourier(emb_size, emb_scale, emb_learnable)
        self.embedder_rel_pos = _Fourier(10, 32, True)
        self.mlp_col_neighbor = _ColNeighbor(c_dim, 20, hidden)
        if exposure:
            self.mlp_exposure = _Exposure(cfg['model']['exposure_dim'], hidden)
        e_in = emb_size * (2 if emb_concat else 1)
        self.pts_linears = nn.ModuleList(
            [_Dense(e_in, hidden)] + [_Dense(hidden, hidden) if i != 2 else _Dense(hidden + e_in, hidden)
                                      for i in range(4)])
        self.output_linear = _Dense(hidden, out_dim, out_act)


class PointDecoders(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg['model'].get('use_view_direction', False):
            raise NotImplementedError("use_view_direction=True is off in every shipped config (point_slam.yaml:15)")
        self.geo_decoder = _Decoder(cfg, 32, 93, 25, False, True, 1, "relu", False)
        self.color_decoder = _Decoder(cfg, 128, 20, 32, True, False, 3, "linear",
                                      cfg['model']['encode_exposure'])

    def load_reference_state(self, tensors: dict):
        """tensors: reference state_dict (+ optional 'color_decoder.embedder._B')."""
        sd = {k: v for k, v in tensors.items() if k != "color_decoder.embedder._B"}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        if unexpected:
            raise KeyError(f"unexpected keys: {unexpected}")
        if "color_decoder.embedder._B" in tensors:
            self.color_decoder.embedder._B = tensors["color_decoder.embedder._B"].clone()
        return self

    def to(self, *a, **k):
        r = super().to(*a, **k)
        r.color_decoder.embedder._B = r.color_decoder.embedder._B.to(*a, **k)
        return r
