"""Caller-side harness: the pieces of DQO-MAP's mapping loop that sit directly around the rasteriser, restated so tests and
bench.py drive the operator exactly the way the reference does (the reference's own callers cannot be imported: they
need CUDA at import time, open3d, pytorch3d — SURVEY.md §8c).

  GaussianParams      <- SLAM/gaussian_pointcloud.py:331-378 (six Adam groups), :723-841 (activations: exp / sigmoid /
                         normalize, shs = cat(f_dc, f_rest))
  render()            <- SLAM/render.py:134-214 (settings, default all-ones tile mask, output dict keys)
  mapping_loss()      <- SLAM/multiprocess/mapper.py:836-875 (0.8 L1 colour + 1.0 depth L1 on valid pixels + 0.2 (1-SSIM)
                         only when no render_mask is given, B14)
  update_geometry_scales() <- SLAM/gaussian_pointcloud.py:519-570 (knn -> scale init)
torch is used here for device memory, autograd plumbing and the optimiser; the rasteriser / knn are the HIP operators.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from diff_gaussian_rasterization_depth import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_gated

# configs/base.yaml:78-86
COLOR_WEIGHT, DEPTH_WEIGHT, SSIM_WEIGHT = 0.8, 1.0, 0.2
LRS = dict(xyz=0.001, f_dc=0.0005, f_rest=0.0005 / 20.0, opacity=0.0, scaling=0.004, rotation=0.001)


class GaussianParams:
    """Raw (pre-activation) parameters as the reference stores them."""

    def __init__(self, scene, device):
        t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=device)
        self._xyz = t(scene["xyz"]).requires_grad_(True)
        shs = t(scene["shs"])
        self._features_dc = shs[:, :1, :].contiguous().requires_grad_(True)
        self._features_rest = shs[:, 1:, :].contiguous().requires_grad_(True)
        op = t(scene["opacity"]).clamp(1e-4, 1 - 1e-4)
        self._opacity = torch.log(op / (1 - op)).requires_grad_(True)       # inverse_sigmoid
        self._scaling = torch.log(t(scene["scales"])).requires_grad_(True)
        self._rotation = t(scene["rotations"]).clone().requires_grad_(True)
        self.normal = t(scene["normals"]) if "normals" in scene else None
        self.obj_id = torch.tensor(scene["obj_id"], device=device) if "obj_id" in scene else None

    def param_groups(self):
        # gaussian_pointcloud.py:338-370 (order and names)
        return [dict(params=[self._xyz], lr=LRS["xyz"], name="xyz"),
                dict(params=[self._features_dc], lr=LRS["f_dc"], name="f_dc"),
                dict(params=[self._features_rest], lr=LRS["f_rest"], name="f_rest"),
                dict(params=[self._opacity], lr=LRS["opacity"], name="opacity"),
                dict(params=[self._scaling], lr=LRS["scaling"], name="scaling"),
                dict(params=[self._rotation], lr=LRS["rotation"], name="rotation")]

    def init_stat(self):
        """`history_stat` of Mapping.local_optimize (mapper.py:533-545): detached copies of the raw parameters at the start of a
        mapping call; loss_update's attach loss pulls towards them."""
        return dict(opacity=self._opacity.detach().clone(), xyz=self._xyz.detach().clone(), scaling=self._scaling.detach().clone(),
                    rotation_raw=self._rotation.detach().clone())

    def activated(self):
        """gaussian_pointcloud.py:732-733, 746-747, 815-822: what the op sees."""
        return dict(xyz=self._xyz, opacity=torch.sigmoid(self._opacity), scales=torch.exp(self._scaling),
                    rotations=F.normalize(self._rotation), shs=torch.cat([self._features_dc, self._features_rest], dim=1),
                    normal=self.normal)


def make_settings(cam, device, sh_degree=3, bg=(0.0, 0.0, 0.0), opaque_threshold=0.6, normal_threshold_deg=60.0,
                  depth_threshold=1.0, color_sigma=3.0):
    """SLAM/render.py:142-162 with the defaults of configs/base.yaml:65-68."""
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=device)
    return GaussianRasterizationSettings(
        image_height=int(cam.H), image_width=int(cam.W), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=t(bg), scale_modifier=1.0,
        viewmatrix=t(cam.world_view_transform), projmatrix=t(cam.full_proj_transform), sh_degree=sh_degree,
        campos=t(cam.camera_center), opaque_threshold=opaque_threshold, depth_threshold=depth_threshold,
        normal_threshold=float(np.cos(np.deg2rad(normal_threshold_deg))), color_sigma=color_sigma, prefiltered=False, debug=False,
        cx=cam.cx, cy=cam.cy, T_threshold=0.0001)


def render(settings, gaussian_data, tile_mask=None, object_gate=None):
    """SLAM/render.py:134-272 (`Renderer.render`): default all-ones int32 tile mask (:178-185), the op's 9-tuple unpacked into the
    reference's dict keys (:213-222, 269-270), the per-pixel normal gathered through the depth hit index (:208-212; ids > -1, so
    pixels of never-rendered tiles alias Gaussian 0, quirk B7), and the optional second / third pass with `semantics_color` /
    `instance` as precomputed colours (:224-266; `semantic_seg` / `instance` are None without them).  `normal` is only gathered
    when the caller supplies per-Gaussian normals (the reference always does)."""
    dev = gaussian_data["xyz"].device
    if tile_mask is None:
        tile_mask = torch.ones(((settings.image_height + 15) // 16, (settings.image_width + 15) // 16), dtype=torch.int32, device=dev)
    rasterizer = GaussianRasterizer(raster_settings=settings)
    normal = gaussian_data.get("normal")
    geo = dict(means3D=gaussian_data["xyz"], opacities=gaussian_data["opacity"], scales=gaussian_data["scales"],
               rotations=gaussian_data["rotations"], cov3D_precomp=None, normal_w=normal, tile_mask=tile_mask)
    if object_gate is None:
        r = rasterizer(shs=gaussian_data["shs"], colors_precomp=None, **geo)
    else:
        e = torch.Tensor([])
        r = rasterize_gaussians_gated(gaussian_data["xyz"], gaussian_data["shs"], e, gaussian_data["opacity"], gaussian_data["scales"],
                                      gaussian_data["rotations"], e, tile_mask, settings, object_gate[0], object_gate[1],
                                      object_gate[2] if len(object_gate) > 2 else None)
    out = {"render": r[0], "depth": r[1], "color_index_map": r[2], "depth_index_map": r[3], "color_hit_weight": r[4],
           "depth_hit_weight": r[5], "T_map": r[6], "n_touched": r[7], "radii": r[8]}
    if normal is not None:
        rn = torch.zeros_like(r[0])
        idx = r[3]
        rn[:, idx[0] > -1] = normal[idx[idx > -1].long()].permute(1, 0)
        out["normal"] = rn
    for src, key in (("semantics_color", "semantic_seg"), ("instance", "instance")):
        cp = gaussian_data.get(src)
        out[key] = rasterizer(shs=None, colors_precomp=cp, **geo)[0] if (cp is not None and cp.numel() > 1) else None
    return out


def perturbed_target(full, settings, device, seed):
    """The synthetic ground truth of a BASELINE configuration ("target image = render of a perturbed copy, so gradients
    are non-trivial") and the per-object screen masks of the per-object job (§8e): render of the map with its centres moved by
    N(0, 4 mm) and its DC colours by N(0, 0.15) (numpy default_rng(seed)); a pixel belongs to the object of the Gaussian that fixes
    its depth in that render, -1 where nothing does.  bench.py and the full-size parity tests build their problem with this ONE
    function.  Returns dict(gt_color [3,H,W], gt_depth [1,H,W], pix_obj int32 [H,W], radii int32 [P]) of GPU tensors."""
    P = full["xyz"].shape[0]
    rng = np.random.default_rng(seed)
    pert = dict(full)
    pert["xyz"] = (full["xyz"] + rng.normal(0, 0.004, full["xyz"].shape)).astype(np.float32)
    pert["shs"] = full["shs"].copy()
    pert["shs"][:, 0, :] += rng.normal(0, 0.15, (P, 3)).astype(np.float32)
    with torch.no_grad():
        # a forward-only render: in the op's 'lazy' / 'deferred' modes nothing behind it would look at the frame's header, and a frame
        # that outgrew the sizes carried over from earlier calls (a camera unlike any before) is a background image — check, and render
        # again with the sizes the flagged frame has raised (why 'exact' is the op's default)
        import diff_gaussian_rasterization_depth as dgr
        act = GaussianParams(pert, device).activated()
        for attempt in range(3):
            tgt = render(settings, act)
            try:
                dgr.verify_pending()
                break
            except RuntimeError:
                if attempt == 2:
                    raise
        hit = tgt["depth_index_map"][0]
        obj_id = torch.tensor(np.asarray(full["obj_id"]), device=device)
        pix_obj = obj_id[hit.long().clamp(min=0)]
        pix_obj[hit < 0] = -1
        return dict(gt_color=tgt["render"].clone(), gt_depth=tgt["depth"].clone(), pix_obj=pix_obj.to(torch.int32).contiguous(),
                    radii=tgt["radii"].clone())


def render_obj(settings, gaussian_data, tile_mask=None):
    """SLAM/render.py:61-132 (`Renderer.render_obj`, the per-object ellipsoid view; dead under the shipped MODE = 1, F3): the same
    rasteriser call as render() with `obj_color` [P, 3] as precomputed colours — no SH, no normals — the default all-ones int32 tile
    mask (:105-112), and only the colour image of the op's 9-tuple handed back, as `render_obj` (:126-130)."""
    dev = gaussian_data["xyz"].device
    if tile_mask is None:
        tile_mask = torch.ones(((settings.image_height + 15) // 16, (settings.image_width + 15) // 16), dtype=torch.int32, device=dev)
    rasterizer = GaussianRasterizer(raster_settings=settings)
    r = rasterizer(means3D=gaussian_data["xyz"], opacities=gaussian_data["opacity"], shs=None, colors_precomp=gaussian_data["obj_color"],
                   scales=gaussian_data["scales"], rotations=gaussian_data["rotations"], cov3D_precomp=None, normal_w=None, tile_mask=tile_mask)
    return {"render_obj": r[0]}


def _gaussian_window(window_size=11, sigma=1.5, channel=3, device="cpu"):
    # utils/loss_utils.py:41-58
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous().to(device)


def ssim(img1, img2, window_size=11):
    # utils/loss_utils.py:60-100 (size_average=True)
    channel = img1.size(-3)
    window = _gaussian_window(window_size, 1.5, channel, img1.device)
    if img1.dim() == 3:
        img1, img2 = img1[None], img2[None]
    mu1 = F.conv2d(img1, window, padding=window_size // 2, groups=channel)
    mu2 = F.conv2d(img2, window, padding=window_size // 2, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=window_size // 2, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=window_size // 2, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=window_size // 2, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()


def attach_loss(params, init_stat):
    """mapper.py:812-829: 1000 * (mse(_scaling[a], s0[a]) + mse(_xyz[a], x0[a]) + mse(_rotation[a], q0[a])) over the Gaussians
    a whose opacity at the start of the mapping call was below 0.9; 0 when there are none.  Written as masked sums / counts: the
    value of the reference's boolean-index form (up to fp32 summation order) without its gather / scatter launches."""
    a = (torch.sigmoid(init_stat["opacity"]) < 0.9).reshape(-1)
    n = a.sum()
    if int(n.item()) == 0:
        return params._xyz.new_zeros(())
    w = a[:, None].to(params._xyz.dtype)
    mse = lambda p, p0: (((p - p0) ** 2) * w).sum() / (n * p.shape[1])
    return 1000 * (mse(params._scaling, init_stat["scaling"]) + mse(params._xyz, init_stat["xyz"]) +
                   mse(params._rotation, init_stat["rotation_raw"]))


def mapping_loss(out, gt_color, gt_depth, render_mask=None, add_depth_thres=0.1):
    """mapper.py:836-875 (normal_weight = 0 in every shipped config).  gt_color [3,H,W], gt_depth [1,H,W], render_mask [H,W]
    bool or None.  The attach term of the same function (mapper.py:812-829) is attach_loss(): the reference keeps it out of the
    reported total and adds it only for the backward, `(loss + attach_loss).backward()` (:905)."""
    image, depth, depth_index = out["render"], out["depth"], out["depth_index_map"]
    ssim_loss = image.new_zeros(())
    if render_mask is None:
        ssim_loss = 1 - ssim(image, gt_color)
        m3 = None
    else:
        m3 = render_mask.bool()
    # masked means written as sum(|x| * mask) / count(mask): same value as the reference's x[mask].mean() (up to fp32
    # summation order) without boolean-index gathers, whose backward costs ~40 tiny sort/merge launches per iteration
    if m3 is None:
        color_loss = torch.abs(image - gt_color).mean()
    else:
        color_loss = (torch.abs(image - gt_color) * m3[None]).sum() / (3.0 * m3.sum().clamp(min=1))
    depth_error = depth - gt_depth
    valid = (depth_index != -1) & (gt_depth > 0) & (depth_error < add_depth_thres)
    if m3 is not None:
        valid = valid & m3[None]
    depth_loss = (torch.abs(depth_error) * valid).sum() / valid.sum().clamp(min=1)
    total = DEPTH_WEIGHT * depth_loss + COLOR_WEIGHT * color_loss + SSIM_WEIGHT * ssim_loss
    return total, dict(total_loss=total.detach(), depth_loss=depth_loss.detach(), color_loss=color_loss.detach(),
                       ssim_loss=ssim_loss.detach())


def per_object_loss(out, gt_color, gt_depth, pixel_object, render_mask=None, add_depth_thres=0.1, n_objects=64):
    """The loss of the per-object job:  L = sum_k L_k,  L_k = mapping_loss's masked colour / depth terms evaluated on
    object k's pixels alone (pixel_object == k, inside render_mask) and normalised by ITS OWN pixel counts — an object's loss and
    gradients do not depend on which other objects are rendered with it, so shards of one map add up to the unsharded job.  Eager torch
    statement of what DqoLossTap.per_object computes inside the blend kernels.  Returns (total, parts)."""
    image, depth, depth_index = out["render"], out["depth"], out["depth_index_map"]
    po = pixel_object.reshape(-1).long()
    m = po >= 0
    if render_mask is not None:
        m = m & render_mask.reshape(-1).bool()
    ids = po.clamp(min=0)
    zeros = lambda: torch.zeros(n_objects, dtype=image.dtype, device=image.device)
    e = torch.abs(image - gt_color).sum(0).reshape(-1) * m
    s_c = zeros().index_add(0, ids, e)
    n_c = zeros().index_add(0, ids, m.to(image.dtype))
    color_loss = (s_c / (3.0 * n_c.clamp(min=1))).sum()
    err = (depth - gt_depth).reshape(-1)
    valid = m & (depth_index.reshape(-1) != -1) & (gt_depth.reshape(-1) > 0) & (err < add_depth_thres)
    s_d = zeros().index_add(0, ids, torch.abs(err) * valid)
    n_d = zeros().index_add(0, ids, valid.to(image.dtype))
    depth_loss = (s_d / n_d.clamp(min=1)).sum()
    total = DEPTH_WEIGHT * depth_loss + COLOR_WEIGHT * color_loss
    return total, dict(total_loss=total.detach(), depth_loss=depth_loss.detach(), color_loss=color_loss.detach(),
                       ssim_loss=image.new_zeros(()))


def make_optimizer(params):
    # mapper.py:548: torch.optim.Adam(l, lr=0.0, eps=1e-15), rebuilt at every keyframe (B13)
    try:
        return torch.optim.Adam(params.param_groups(), lr=0.0, eps=1e-15, fused=True)
    except (TypeError, RuntimeError):
        return torch.optim.Adam(params.param_groups(), lr=0.0, eps=1e-15)


def update_geometry_scales(xyz, radius, extra_xyz=None, extra_radius=None, min_radius=0.001, max_radius=0.05,
                           xyz_factor=(1.0, 1.0, 0.1), scale_factor=1.0):
    """gaussian_pointcloud.py:519-570 without the container bookkeeping: returns (log_scales[P,3], invalid_mask[P])."""
    from simple_knn._C import distCUDA2
    points_num = xyz.size(0)
    if extra_xyz is not None and extra_xyz.numel() > 0:
        lo, hi = xyz.min(0).values, xyz.max(0).values  # bbox_filter
        inb = ((extra_xyz >= lo) & (extra_xyz <= hi)).all(dim=1)
        extra_xyz, extra_radius = extra_xyz[inb], extra_radius[inb]
        total_xyz, total_radius = torch.cat([xyz, extra_xyz]), torch.cat([radius, extra_radius])
    else:
        total_xyz, total_radius = xyz, radius
    _, knn_indices = distCUDA2(total_xyz.float().contiguous())
    knn_indices = knn_indices[:points_num].long()
    dists, invalid = [], torch.zeros(points_num, dtype=torch.bool, device=xyz.device)
    for k in range(3):
        d = torch.norm(xyz - total_xyz[knn_indices[:, k]], p=2, dim=1) - 3 * total_radius[knn_indices[:, k]]
        invalid |= d < 0
        dists.append(d)
    dist2 = (dists[0] ** 2 + dists[1] ** 2 + dists[2] ** 2) / 3
    scales = torch.clip(torch.sqrt(dist2), min=min_radius, max=max_radius)
    scales = scales[..., None].repeat(1, 3)
    factor = scale_factor * scales * torch.tensor(xyz_factor, device=xyz.device)
    return torch.log(factor), invalid
