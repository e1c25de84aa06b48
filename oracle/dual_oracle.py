"""CPU ORACLE for BASELINE.json configs[3] (cfg 4) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

"Static + dynamic EmerNeRF-style dual field (two hash grids + flow MLP)".

Parity status: **UNPINNED**.  The reference has no dynamic / flow field anywhere (the
only EmerNeRF-derived code in /root/reference is the DINO feature extractor, ns/scripts/datasets/extract_dino_features.py:2),
so there is no reference artefact this file could be checked against.  It is the build's own DEFINITION of the cfg-4 model,
written first, as plain torch-CPU code whose gradients come from autograd, and it is what the HIP kernels of
presight_amd/csrc/dynamic.hip are checked against.  What IS pinned: every building block it re-uses from
oracle/nerf_oracle.py (static field, samplers, renderers, losses), and the 3-D restriction of the 4-D hash grid
(tests/test_dual_oracle.py: a 4-D table queried at integer times equals the pinned 3-D encode of the matching 3-D table).

Model definition (conventions follow the static stack of ns/fields/PreSight/ingp_field.py:168-267 wherever one exists):

  static branch   = the reference's iNGPField, unchanged: (sigma_s, rgb_s, sem_s) per sample.
  dynamic branch  : u = the static branch's normalised + contracted position in [0,1]^3 (ingp_field.py:169-177), t = the ray's
                    normalised timestamp in [0,1];
      e0      = H4(u, t)                          4-D multiresolution hash grid (hash_encode4 below)
      flow    = flow_scale * MLP_flow(e0)         [.., 6] = forward flow (3) | backward flow (3), in units of u
      feat    = (e0 + H4(u + flow_f, t + dt) + H4(u + flow_b, t - dt)) / 3        EmerNeRF's temporal aggregation
      [raw_d | geo15 | sem64] = MLP_base(feat);  sigma_d = trunc_exp(raw_d) * selector
      sem_d = MLP_sem(sem64);  rgb_d = sigmoid(MLP_rgb([SH16((d+1)/2) | geo15 | appearance]))
                    (same three MLP shapes as the static field: ingp_field.py:130-161)
  blend (per sample)   sigma = sigma_s + sigma_d,  w_d = sigma_d / max(sigma, 1e-6),
                       rgb = rgb_s + w_d (rgb_d - rgb_s),  sem = sem_s + w_d (sem_d - sem_s)
                    (the density-weighted mixture of EmerNeRF written as a lerp, so that sigma_d == 0 reproduces the static
                    model bit for bit)
  then RaySamples.get_weights / renderers / sky / losses exactly as the static model, plus EmerNeRF's dynamic-density
  regulariser  dynamic_reg_mult * mean(sigma_d).
  The flow field has no loss of its own: it is learned through the aggregation ("emergent flow").  The proposal networks stay
  static (they are supervised by the blended weights through the interlevel loss).

4-D hash grid H4 = the torch-path 3-D grid of ns/field_components/encodings.py:324-384 with one more axis: per level
scaled = x * scalings[l] on all four axes, ceil / floor corners, offset = scaled - floor(scaled); index = (x*1 ^ y*2654435761 ^
z*805459861 ^ t*3674653429) mod T + l*T (the fourth prime is Instant-NGP's); the 8 spatial corners of each time corner are
blended in the reference's order (x, then y, then z), the two results are blended along t.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import nerf_oracle as O

PRIME_T = 3674653429


def hash_index4(ix: Tensor, iy: Tensor, iz: Tensor, it: Tensor, level_offset: Tensor, table_size: int) -> Tensor:
    h = (ix.to(torch.int64) ^ (iy.to(torch.int64) * O.PRIME_Y) ^ (iz.to(torch.int64) * O.PRIME_Z) ^ (it.to(torch.int64) * PRIME_T))
    return h % table_size + level_offset


def hash_encode4(x: Tensor, table: Tensor, scalings: Tensor, log2_T: int, return_indices: bool = False):
    """x [N,4] (u_x, u_y, u_z, t) -> [N, L*F]; differentiable w.r.t. the table AND w.r.t. x (through the offsets:
    d offset / d x = scalings[l], floor / ceil carry no gradient)."""
    assert x.shape[-1] == 4
    T = 1 << log2_T
    L = scalings.numel()
    scaled = x[:, None, :] * scalings.view(1, L, 1)  # [N,L,4]
    c = torch.ceil(scaled).detach().to(torch.int32)
    f = torch.floor(scaled).detach().to(torch.int32)
    o = scaled - torch.floor(scaled).detach()
    off = (torch.arange(L, dtype=torch.int64) * T).view(1, L)
    pick = {"c": c, "f": f}
    ox, oy, oz, ot = o[..., 0:1], o[..., 1:2], o[..., 2:3], o[..., 3:4]
    outs, all_idx = [], []
    for tk in ("c", "f"):
        idx = [hash_index4(pick[k[0]][..., 0], pick[k[1]][..., 1], pick[k[2]][..., 2], pick[tk][..., 3], off, T) for k in O._CORNERS]
        all_idx += idx
        v = [table[i] for i in idx]
        f03 = v[0] * ox + v[3] * (1 - ox)
        f12 = v[1] * ox + v[2] * (1 - ox)
        f56 = v[5] * ox + v[6] * (1 - ox)
        f47 = v[4] * ox + v[7] * (1 - ox)
        f0312 = f03 * oy + f12 * (1 - oy)
        f4756 = f47 * oy + f56 * (1 - oy)
        outs.append(f0312 * oz + f4756 * (1 - oz))
    out = (outs[0] * ot + outs[1] * (1 - ot)).flatten(-2)
    if return_indices:
        return out, torch.stack(all_idx, dim=-1)  # [N,L,16]: the 8 spatial corners at ceil(t), then at floor(t)
    return out


def dyn_defaults() -> dict:
    """cfg 4 at full size: static = cfg-2 field, dynamic grid L8 F4 T2^19 over (x,y,z,t) at resolutions 16..512, 64-wide MLPs"""
    return dict(num_levels=8, features_per_level=4, log2_hashmap_size=19, base_res=16, max_res=512, hidden_dim=64,
                hidden_dim_color=64, flow_hidden_dim=64, flow_scale=0.05, time_step=1.0 / 240.0, dynamic_reg_mult=0.01,
                geo_feat_dim=15, semantic_dim=64)


def dual_config(tiny: bool = True, levels: int = 2, feats: int = 2) -> dict:
    cfg = O.tiny_config() if tiny else O.default_config()
    d = dyn_defaults()
    if tiny:
        d.update(num_levels=levels, features_per_level=feats, log2_hashmap_size=12, max_res=64, hidden_dim=32, hidden_dim_color=32,
                 flow_hidden_dim=32, time_step=1.0 / 12.0)
    cfg["dynamic"] = d
    return cfg


def make_dual_params(cfg: dict, seed: int = 42, table_scale: float = 1e-3) -> Dict[str, Tensor]:
    """static parameters of O.make_params + the dynamic field's (`dynamic_field.*`, same naming scheme as field.fields.0.*)"""
    P = O.make_params(cfg, seed=seed, table_scale=table_scale)
    gen = torch.Generator().manual_seed(seed + 1000)
    d = cfg["dynamic"]
    app = cfg["appearance_embed_dim"] + cfg["video_embed_dim"]
    T = 1 << d["log2_hashmap_size"]
    nin = d["num_levels"] * d["features_per_level"]
    P["dynamic_field.encoding.hash_table"] = (torch.rand(T * d["num_levels"], d["features_per_level"], generator=gen) * 2 - 1) * table_scale
    O._add_mlp(P, gen, "dynamic_field.mlp_base_mlp", [nin, d["hidden_dim"], 1 + d["geo_feat_dim"] + d["semantic_dim"]])
    O._add_mlp(P, gen, "dynamic_field.semantic_head", [d["semantic_dim"], 64, 64, d["semantic_dim"]])
    O._add_mlp(P, gen, "dynamic_field.rgb_head", [16 + d["geo_feat_dim"] + app, d["hidden_dim_color"], d["hidden_dim_color"], 3])
    O._add_mlp(P, gen, "dynamic_field.flow_head", [nin, d["flow_hidden_dim"], d["flow_hidden_dim"], 6])
    return P


def ray_times(scene: dict, ray_indices: Tensor) -> Tensor:
    """normalised timestamp of a ray = frame index of its camera / (frames - 1); cameras are stored frame-major, 6 per frame
    (O.make_scene)"""
    C = scene["c2w"].shape[0]
    n_frames = max(1, C // 6)
    return (ray_indices[:, 0] // 6).to(torch.float32) / float(max(1, n_frames - 1))


def dynamic_features(P, cfg, u: Tensor, t: Tensor, return_parts: bool = False):
    """u [M,3] (normalised, contracted, masked), t [M] -> aggregated dynamic features [M, L*F]"""
    d = cfg["dynamic"]
    sc = O.hash_scalings(d["num_levels"], d["base_res"], d["max_res"])
    tab = P["dynamic_field.encoding.hash_table"]
    x0 = torch.cat([u, t[:, None]], -1)
    e0 = hash_encode4(x0, tab, sc, d["log2_hashmap_size"])
    flow = d["flow_scale"] * O.mlp_forward(e0, O._mlp_layers(P, "dynamic_field.flow_head"))
    dt = d["time_step"]
    xf = torch.cat([u + flow[:, 0:3], (t + dt)[:, None]], -1)
    xb = torch.cat([u + flow[:, 3:6], (t - dt)[:, None]], -1)
    ef = hash_encode4(xf, tab, sc, d["log2_hashmap_size"])
    eb = hash_encode4(xb, tab, sc, d["log2_hashmap_size"])
    feat = (e0 + ef + eb) / 3.0
    if return_parts:
        return feat, dict(e0=e0, flow=flow, xf=xf, xb=xb, ef=ef, eb=eb)
    return feat


def dynamic_eval(P, cfg, pos: Tensor, t: Tensor, dirs: Tensor, app: Optional[Tensor], aabb: Tensor):
    """-> (sigma_d [M], rgb_d [M,3], sem_d [M,64])"""
    d = cfg["dynamic"]
    u, sel = O.normalize_contract(pos, aabb)
    feat = dynamic_features(P, cfg, u, t)
    h = O.mlp_forward(feat, O._mlp_layers(P, "dynamic_field.mlp_base_mlp"))
    sigma = O.trunc_exp(h[:, 0]) * sel
    emb = h[:, 1:]
    geo, sem_in = emb[:, : d["geo_feat_dim"]], emb[:, d["geo_feat_dim"]:]
    sem = O.mlp_forward(sem_in, O._mlp_layers(P, "dynamic_field.semantic_head"))
    parts = [O.sh4_of_direction(dirs), geo] + ([app] if app is not None else [])
    rgb = O.mlp_forward(torch.cat(parts, -1), O._mlp_layers(P, "dynamic_field.rgb_head"), out_act="sigmoid")
    return sigma, rgb, sem


def blend(sigma_s, rgb_s, sem_s, sigma_d, rgb_d, sem_d, eps: float = 1e-6) -> Tuple[Tensor, Tensor, Tensor]:
    sigma = sigma_s + sigma_d
    wd = (sigma_d / torch.clamp(sigma, min=eps))[:, None]
    return sigma, rgb_s + wd * (rgb_d - rgb_s), sem_s + wd * (sem_d - sem_s)


def dynamic_aabb(scene: dict) -> Tensor:
    """the box the dynamic field is normalised by: the sub-field's AABB (K = 1) / the union of the K sub-field boxes of a routed tile
    (the static branch is routed like the reference's iNGPFieldMS, the dynamic branch is ONE field over the whole tile)"""
    b = scene["aabbs"]
    return b[0] if b.shape[0] == 1 else torch.stack([b[:, 0].min(0).values, b[:, 1].max(0).values])


def dual_model_forward(P, cfg, scene, batch, training: bool = True, anneal: float = 1.0):
    times = batch["times"] if "times" in batch else ray_times(scene, batch["ray_indices"])

    def override(static_eval, pos, dir_s, app_s, R, S):
        sigma_s, rgb_s, sem_s = static_eval()
        t_s = times[:, None].expand(R, S).reshape(-1)
        sigma_d, rgb_d, sem_d = dynamic_eval(P, cfg, pos, t_s, dir_s, app_s, dynamic_aabb(scene))
        sigma, rgb, sem = blend(sigma_s.view(-1), rgb_s, sem_s, sigma_d, rgb_d, sem_d)
        return sigma, rgb, sem, dict(dynamic_density=sigma_d, static_density=sigma_s.view(-1))

    return O.model_forward(P, cfg, scene, batch, training=training, anneal=anneal, main_override=override)


def dual_loss_dict(out, batch, cfg) -> Dict[str, Tensor]:
    L = O.loss_dict(out, batch, cfg)
    L["dynamic_reg_loss"] = cfg["dynamic"]["dynamic_reg_mult"] * out["dynamic_density"].mean()
    return L


def dual_train_step(P, cfg, scene, batch, anneal: float = 1.0):
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = dual_model_forward(Pg, cfg, scene, batch, training=True, anneal=anneal)
    L = dual_loss_dict(out, batch, cfg)
    sum(L.values()).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    return L, out, grads
