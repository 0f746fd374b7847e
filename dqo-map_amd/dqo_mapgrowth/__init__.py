"""Map-growth geometry of DQO-MAP's mapper on MI355X (second half).

    knn_points_k3        pytorch3d.ops.knn_points(p1[None], p2[None], K=3, norm=2) as the reference calls it
                         (SLAM/multiprocess/mapper.py:1366-1372): exact squared distances, ascending, + indices into p2
    bbox_filter          SLAM/utils.py:801-808
    temp_points_filter_mask   the decision of Mapping.temp_points_filter (mapper.py:1351-1380): which new ("temp") points
                         fall inside an existing unstable Gaussian (nearest-3 distance < 0.6 x its radius)
    update_geometry_scales    the scale initialisation of GaussianPointCloud.update_geometry
                         (SLAM/gaussian_pointcloud.py:519-570) on top of simple_knn's distCUDA2

    *_per_object         the same two decisions for the per-object job: a candidate is judged against the Gaussians of ITS
                         OWN object only, so that a sharded map grows exactly like the unsharded one (not a reference feature)

    temp_points_attach_indices   the decision of Mapping.temp_points_attach (mapper.py:1384-1430): which new points lie on a stable
                         Gaussian's plane in the current frame (they get opacity 0.1 and thereby become the members of the attach
                         loss, mapper.py:812-829) — gathers over the stable render's color_index_map, no kernel of its own

The nearest-neighbour search is libdqoraster.so's dqo_knn3_query (Morton-sorted, box-pruned, wave-uniform candidate loads —
csrc/knn.hip); everything else is a handful of element-wise torch ops exactly as in the reference.  GPU only.
"""
import torch

import _dqo_native as N
from simple_knn._C import distCUDA2


def knn_points_k3(p1, p2, max_dist=None, groups=None, group_box=None, int32_idx=False):
    """(dists [Q, 3] squared L2 ascending, idx [Q, 3] int64 into p2).  Fewer than 3 references: FLT_MAX / -1 in the tail.
    max_dist (not a pytorch3d argument): only references closer than that count (dqo_knn3_query_within).
    groups = (g1 [Q], g2 [R]) int32 ids in [0, 64) (any other value: the point belongs to no group): a reference only counts for a
    query of the same group; group_box [64, 6] (lo xyz, hi xyz): ... and only if it lies strictly inside its group's box
    (dqo_knn3_query_grouped)."""
    N.require_gpu(p1, p2)
    if not (p1.is_cuda and p2.is_cuda):
        raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
    q = p1.float().contiguous()
    r = p2.float().contiguous()
    Q, R = q.shape[0], r.shape[0]
    d = torch.empty((Q, 3), dtype=torch.float32, device=q.device)
    i = torch.empty((Q, 3), dtype=torch.int32, device=q.device)
    if Q == 0:
        return d, (i if int32_idx else i.long())
    if R == 0:
        raise RuntimeError("knn_points_k3: empty reference set")
    lib = N.lib()
    ws = torch.empty((lib.dqo_knn3_query_workspace_bytes(Q, R),), dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        if groups is not None:
            g1 = groups[0].to(torch.int32).contiguous()
            g2 = groups[1].to(torch.int32).contiguous()
            assert g1.shape[0] == Q and g2.shape[0] == R
            gb = None if group_box is None else group_box.float().contiguous()
            assert gb is None or tuple(gb.shape) == (64, 6)
            N.check(lib.dqo_knn3_query_grouped(Q, N.ptr(q), N.ptr(g1), R, N.ptr(r), N.ptr(g2), None if gb is None else N.ptr(gb),
                                               float(max_dist) if max_dist is not None else 3.0e38, N.ptr(d), N.ptr(i), N.ptr(ws), ws.numel(),
                                               N.current_stream()))
        elif max_dist is None:
            N.check(lib.dqo_knn3_query(Q, N.ptr(q), R, N.ptr(r), N.ptr(d), N.ptr(i), N.ptr(ws), ws.numel(), N.current_stream()))
        else:
            N.check(lib.dqo_knn3_query_within(Q, N.ptr(q), R, N.ptr(r), float(max_dist), N.ptr(d), N.ptr(i), N.ptr(ws), ws.numel(),
                                              N.current_stream()))
    return d, (i if int32_idx else i.long())  # (int32_idx: the kernel's own index type, for callers that hand it to another kernel)


def bbox_filter(local_xyz, total_xyz, padding=0.05):
    """SLAM/utils.py:801-808: mask of the total_xyz points strictly inside the padded bounding box of local_xyz."""
    local_min = local_xyz.min(dim=0)[0] - padding
    local_max = local_xyz.max(dim=0)[0] + padding
    return (total_xyz > local_min).all(dim=-1) & (total_xyz < local_max).all(dim=-1)


def temp_points_filter_mask(temp_xyz, exist_xyz, exist_radius, topk=3):
    """mapper.py:1351-1380: True for the temp points to delete.  Returns None where the reference returns early."""
    if topk != 3:
        raise ValueError("the kernel is built for K = 3, the only value the reference uses")
    if torch.numel(exist_xyz) > 0 and torch.numel(temp_xyz) > 0:
        inbbox = bbox_filter(temp_xyz, exist_xyz)
        exist_xyz, exist_radius = exist_xyz[inbbox], exist_radius[inbbox]
    if torch.numel(exist_xyz) == 0:
        return None
    nn_dist, nn_idx = knn_points_k3(temp_xyz, exist_xyz)
    valid = nn_idx >= 0  # fewer than 3 existing points in the box
    nn_dist = torch.sqrt(nn_dist)
    corr_radius = exist_radius.reshape(-1)[nn_idx.clamp(min=0)] * 0.6
    return ((nn_dist < corr_radius) & valid).any(dim=-1)


def update_geometry_scales(xyz, radius, extra_xyz, extra_radius, min_radius, max_radius, literal=False):
    """gaussian_pointcloud.py:519-556: (scales [P], invalid_scale_mask [P]) for the P new points `xyz` among the
    existing `extra_xyz`: root-mean-square gap to the 3 nearest neighbours' 3-sigma spheres, clipped.

    literal=True issues the reference's own call — distCUDA2 over the concatenation (new + existing points in the box), i.e. an
    all-pairs 3-NN over up to the whole map of which only the first P rows are used.  The default computes the same three neighbours
    per new point without that waste: its nearest 3 among the other new points (dqo_knn3) and among the existing points
    (dqo_knn3_query), then the 3 nearest of those 6 — the 3 nearest of the union, exactly (equidistant neighbours aside)."""
    n = xyz.shape[0]
    if torch.numel(extra_xyz) > 0:
        inbbox = bbox_filter(xyz, extra_xyz)
        extra_xyz, extra_radius = extra_xyz[inbbox], extra_radius[inbbox]
    if literal:
        total_xyz = torch.cat([xyz, extra_xyz])
        total_radius = torch.cat([radius, extra_radius]).reshape(-1)
        _, knn_indices = distCUDA2(total_xyz.float().cuda())
        knn_indices = knn_indices[:n].long()
        d = [torch.norm(xyz - total_xyz[knn_indices[:, j]], p=2, dim=1) - 3 * total_radius[knn_indices[:, j]] for j in range(3)]
    else:
        xyz = xyz.float().contiguous()
        radius, extra_radius = radius.reshape(-1), extra_radius.reshape(-1)
        inf = torch.full((n, 3), float("inf"), device=xyz.device)
        cand_d, cand_r = [], []
        if n > 1:
            _, i_new = distCUDA2(xyz)
            ok = i_new < n  # (INT_MAX where fewer than 3 other new points exist)
            j = i_new.long().clamp(max=n - 1)
            cand_d.append(torch.where(ok, (xyz[:, None, :] - xyz[j]).pow(2).sum(-1), inf))
            cand_r.append(radius[j])
        if torch.numel(extra_xyz) > 0:
            d2, i_old = knn_points_k3(xyz, extra_xyz)
            ok = i_old >= 0
            cand_d.append(torch.where(ok, d2, inf))
            cand_r.append(extra_radius[i_old.clamp(min=0)])
        cd, cr = torch.cat(cand_d, 1), torch.cat(cand_r, 1)
        top = torch.topk(cd, 3, dim=1, largest=False)
        dist = torch.sqrt(top.values)
        rr = torch.gather(cr, 1, top.indices)
        d = [dist[:, k] - 3 * rr[:, k] for k in range(3)]
    invalid = (d[0] < 0) | (d[1] < 0) | (d[2] < 0)
    scales = torch.sqrt((d[0] ** 2 + d[1] ** 2 + d[2] ** 2) / 3)
    return torch.clip(scales, min=min_radius, max=max_radius), invalid


# ---- the per-object job: every growth decision of a candidate looks at the Gaussians of the candidate's own object only ----
# A shard of the map holds whole objects, so a decision that only looks at the candidate's object is the same on every shard layout:
# the N-rank map grows exactly like the N = 1 map.  The search against the EXISTING map is the group-restricted one
# (dqo_knn3_query_grouped, round 5): object ids compared inside the search, the per-object bounding boxes tested inside it too — one
# search over the map as it is stored (round 4 moved every object to a cell of its own and searched shifted copies of gathered subsets:
# 3.5 ms against 1.5 ms for the search alone on cfg 5, plus a 0.9 ms mask and the gathers per call).  The new points among themselves
# (40 800 of them: distCUDA2, 0.3 ms) still use the cells: every object is moved to a cell of its own on a grid (ids in [0, 64): 4 x 4 x 4
# cells of `cell` metres per axis, powers of two so that the shift is exact for most coordinates), so a point's nearest neighbours are
# its own object's whenever that object has three nearby — checked explicitly, never assumed — and the distances the decisions use are
# recomputed from the unshifted coordinates of the pairs found (the shift only selects neighbours, it never enters a value).
OBJECT_CELL = (16.0, 16.0, 16.0)
# ... and a neighbour further away than this is as good as none for both decisions (radii up to 0.3 m), so the searches stop there
# (dqo_knn3_query_within): a candidate whose object has nothing nearby — common at an object's rim — does not scan half the map for three
# far neighbours that cannot change its scale.  Part of the per-object job's definition: the same on every shard layout.
NEIGHBOUR_REACH = 1.0


_const_cache = {}


def const_tensor(values, device, dtype=torch.float32):
    """A small constant on the device, made once per (values, device): torch.tensor(list, device=cuda) is a synchronous host-to-device
    copy of pageable memory — about a millisecond each on MI355X boxes, and the growth step made four of them per call."""
    import numpy as np
    key = (tuple(np.asarray(values, np.float64).reshape(-1).tolist()), tuple(np.asarray(values).shape), str(device), dtype)
    t = _const_cache.get(key)
    if t is None:
        t = _const_cache[key] = torch.tensor(values, dtype=dtype, device=device)
    return t


def object_offsets(obj, cell=None):
    """[n, 3] float32 translation of every point's object cell (ids in [0, 64))."""
    o = obj.long()
    c = const_tensor(OBJECT_CELL if cell is None else cell, obj.device)
    return torch.stack([o % 4, (o // 4) % 4, o // 16], dim=1).to(torch.float32) * c


def _per_object_boxes(query_xyz, query_obj, padding=0.05, n_objects=64):
    """[n_objects, 6] (lo xyz, hi xyz): the padded bounding box of every object's QUERY points (an object without queries: an empty box)."""
    inf = float("inf")
    idx = query_obj.long()[:, None].expand(-1, 3)
    lo = torch.full((n_objects, 3), inf, device=query_xyz.device).scatter_reduce(0, idx, query_xyz, "amin", include_self=True)
    hi = torch.full((n_objects, 3), -inf, device=query_xyz.device).scatter_reduce(0, idx, query_xyz, "amax", include_self=True)
    return torch.cat([lo - padding, hi + padding], dim=1)


def _per_object_bbox_mask(query_xyz, query_obj, ref_xyz, ref_obj, padding=0.05, n_objects=64):
    """bbox_filter per object: a reference point passes iff it lies strictly inside the padded bounding box of the QUERY points of its
    own object (no query of that object: it does not pass).  (What dqo_knn3_query_grouped tests inside the search; kept as the
    statement the tests compare it with.)"""
    box = _per_object_boxes(query_xyz, query_obj, padding, n_objects)
    ro = ref_obj.long().clamp(0, n_objects - 1)
    ok = (ref_obj >= 0) & (ref_obj < n_objects)
    return ok & (ref_xyz > box[ro, :3]).all(dim=-1) & (ref_xyz < box[ro, 3:]).all(dim=-1)


def temp_points_filter_mask_per_object(temp_xyz, temp_obj, exist_xyz, exist_radius, exist_obj, cell=None, fused=True):
    """temp_points_filter_mask with every candidate judged against the existing Gaussians of its own object: True for the temp points
    that lie within 0.6 x radius of one of the (up to) 3 nearest existing centres of their object.  None: nothing to test against.
    fused: the decision behind the search in one launch (dqo_growth_inside) instead of the torch chain below it — the same bits."""
    if torch.numel(exist_xyz) == 0 or torch.numel(temp_xyz) == 0:
        return None
    nn_d2, nn_idx = knn_points_k3(temp_xyz, exist_xyz, max_dist=NEIGHBOUR_REACH, groups=(temp_obj, exist_obj),
                                  group_box=_per_object_boxes(temp_xyz, temp_obj), int32_idx=fused)
    if fused:
        n = int(temp_xyz.shape[0])
        er = exist_radius.reshape(-1).float().contiguous()
        out = torch.empty((n,), dtype=torch.uint8, device=temp_xyz.device)
        with torch.cuda.device(temp_xyz.device):
            N.check(N.lib().dqo_growth_inside(n, N.ptr(nn_d2), N.ptr(nn_idx), N.ptr(er), N.ptr(out), N.current_stream()))
        return out.bool()
    j = nn_idx.clamp(min=0)
    valid = nn_idx >= 0
    return ((torch.sqrt(nn_d2) < exist_radius.reshape(-1)[j] * 0.6) & valid).any(dim=-1)


def update_geometry_scales_per_object(xyz, obj, radius, extra_xyz, extra_radius, extra_obj, min_radius, max_radius, cell=None, fused=True):
    """update_geometry_scales with every new point's three neighbours taken from its own object (the other new points of the object and
    the object's existing points inside the bounding box of the object's new points).  A point whose object offers fewer than three
    neighbours keeps `inf` in the missing slots: its scale is clipped to max_radius, on every shard layout alike.
    fused: everything behind the two searches in one launch (dqo_growth_scales) instead of the torch chain below — the same bits."""
    n = xyz.shape[0]
    xyz = xyz.float().contiguous()
    radius, extra_radius = radius.reshape(-1), extra_radius.reshape(-1)
    if fused:
        dev = xyz.device
        obj32, rad = obj.to(torch.int32).contiguous(), radius.float().contiguous()
        i_new = d2_old = i_old = er = None
        if n > 1:
            _, i_new = distCUDA2((xyz + object_offsets(obj, cell)).contiguous())
        if torch.numel(extra_xyz) > 0:
            d2_old, i_old = knn_points_k3(xyz, extra_xyz, max_dist=NEIGHBOUR_REACH, groups=(obj, extra_obj), group_box=_per_object_boxes(xyz, obj),
                                          int32_idx=True)
            er = extra_radius.float().contiguous()
        scales = torch.empty((n,), dtype=torch.float32, device=dev)
        invalid = torch.empty((n,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            N.check(N.lib().dqo_growth_scales(n, N.ptr(xyz), N.ptr(obj32), N.ptr(rad), N.ptr(i_new), N.ptr(d2_old), N.ptr(i_old), N.ptr(er),
                                              NEIGHBOUR_REACH * NEIGHBOUR_REACH, float(min_radius), float(max_radius), N.ptr(scales),
                                              N.ptr(invalid), N.current_stream()))
        return scales, invalid.bool()
    inf = torch.full((n, 3), float("inf"), device=xyz.device)
    shifted = (xyz + object_offsets(obj, cell)).contiguous()
    cand_d, cand_r = [inf], [torch.zeros_like(inf)]
    if n > 1:
        _, i_new = distCUDA2(shifted)
        j = i_new.long().clamp(max=n - 1)
        # (the three squares added in a fixed order — torch.sum over a dimension of three adds (0 + 2) + 1 in this build, another build
        # may differ: the kernel and this chain spell the order out)
        df = xyz[:, None, :] - xyz[j]
        d2_new = (df[..., 0] * df[..., 0] + df[..., 1] * df[..., 1]) + df[..., 2] * df[..., 2]
        ok = (i_new < n) & (obj[j] == obj[:, None]) & (d2_new < NEIGHBOUR_REACH * NEIGHBOUR_REACH)
        cand_d.append(torch.where(ok, d2_new, inf))
        cand_r.append(radius[j])
    if torch.numel(extra_xyz) > 0:
        d2_old, i_old = knn_points_k3(xyz, extra_xyz, max_dist=NEIGHBOUR_REACH, groups=(obj, extra_obj), group_box=_per_object_boxes(xyz, obj))
        j = i_old.clamp(min=0)
        cand_d.append(torch.where(i_old >= 0, d2_old, inf))
        cand_r.append(extra_radius[j])
    cd, cr = torch.cat(cand_d, 1), torch.cat(cand_r, 1)
    top = torch.topk(cd, 3, dim=1, largest=False)
    dist = torch.sqrt(top.values)
    rr = torch.gather(cr, 1, top.indices)
    d = [dist[:, k] - 3 * rr[:, k] for k in range(3)]
    invalid = (d[0] < 0) | (d[1] < 0) | (d[2] < 0)
    scales = torch.sqrt((d[0] ** 2 + d[1] ** 2 + d[2] ** 2) / 3)
    return torch.clip(scales, min=min_radius, max=max_radius), invalid


def error_maps(gt_color, gt_depth, render, depth, depth_index, render_mask=None, fused=True):
    """The per-pixel error images of SLAM/multiprocess/mapper.py:1016-1033 that feed accumulate_gaussian_error: (color_err [1,H,W] = sum over
    channels |gt_color - render|, depth_err [1,H,W] = max(gt_depth - depth, 0)), both zero where gt_depth == 0 or outside render_mask
    ([H,W] bool / uint8, optional), depth_err also where the pixel has no depth hit (depth_index == -1).  fused: one launch
    (dqo_error_maps) instead of the torch chain below — the same bits."""
    H, W = int(gt_depth.shape[-2]), int(gt_depth.shape[-1])
    if fused:
        dev = gt_depth.device
        c = lambda a: a.contiguous()
        m = None if render_mask is None else render_mask.to(torch.uint8).contiguous()
        color_err, depth_err = torch.empty((1, H, W), dtype=torch.float32, device=dev), torch.empty((1, H, W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(N.lib().dqo_error_maps(H, W, N.ptr(c(gt_color)), N.ptr(c(gt_depth)), N.ptr(c(render)), N.ptr(c(depth)), N.ptr(c(depth_index)),
                                           N.ptr(m), N.ptr(color_err), N.ptr(depth_err), N.current_stream()))
        return color_err, depth_err
    off = gt_depth == 0
    if render_mask is not None:
        off = off | ~render_mask.bool()[None]
    depth_err = (gt_depth - depth).clamp(min=0)
    depth_err.masked_fill_(off | (depth_index == -1), 0)
    color_err = (gt_color - render).abs().sum(0, keepdim=True)
    color_err.masked_fill_(off, 0)
    return color_err, depth_err


def temp_points_pixels(temp_xyz, w2c, intrinsic, image_width, image_height):
    """scene/cameras.py:207-214 get_uv as mapper.py:1398-1404 uses it: pixel = trunc(K (R x + t) / z) of every point (long [N, 2]: u, v)
    and whether it lies inside the image.  Written out element by element — separate multiplies and adds in a fixed order, K = [[fx, 0,
    cx], [0, fy, cy], [0, 0, 1]] — so that a point's pixel does not depend on the other points of the batch (a matmul's blocking may)
    and equals dqo_attach_pixels' bit for bit (csrc/map_attach.hip follows this function)."""
    x, y, z = temp_xyz[:, 0], temp_xyz[:, 1], temp_xyz[:, 2]
    R, t, K = w2c[:3, :3], w2c[:3, 3], intrinsic
    xc = ((x * R[0, 0] + y * R[0, 1]) + z * R[0, 2]) + t[0]
    yc = ((x * R[1, 0] + y * R[1, 1]) + z * R[1, 2]) + t[1]
    zc = ((x * R[2, 0] + y * R[2, 1]) + z * R[2, 2]) + t[2]
    uf, vf = (xc * K[0, 0] + zc * K[0, 2]) / zc, (yc * K[1, 1] + zc * K[1, 2]) / zc
    inside = (uf > -1) & (uf < image_width) & (vf > -1) & (vf < image_height)  # (trunc toward zero: pixel >= 0 <=> coordinate > -1)
    uv = torch.stack([torch.where(inside, uf, torch.zeros_like(uf)), torch.where(inside, vf, torch.zeros_like(vf))], dim=1).long()
    return uv, inside


def attach_pixels(temp_xyz, viewmatrix, fx, fy, cx, cy, image_width, image_height, pixel_object):
    """dqo_attach_pixels (include/dqo_raster.h): (lin int32 [N] — the candidates' pixels v * W + u, -1 = outside the image;
    sparse pixel_object int32 [H * W]; tile_objects int64 [tiles]) — the sparse object gate of the attach render, in two launches."""
    import _dqo_native as N
    n, dev = int(temp_xyz.shape[0]), temp_xyz.device
    W, H = int(image_width), int(image_height)
    N.require_gpu(temp_xyz, viewmatrix, pixel_object)
    if temp_xyz.dtype != torch.float32 or viewmatrix.dtype != torch.float32 or pixel_object.dtype != torch.int32:
        raise RuntimeError("attach_pixels: float32 points / viewmatrix, int32 pixel_object")
    if pixel_object.numel() != W * H or viewmatrix.numel() != 16:
        raise RuntimeError("attach_pixels: pixel_object must have H x W elements, viewmatrix 16")
    temp_xyz, viewmatrix, pixel_object = temp_xyz.contiguous(), viewmatrix.contiguous(), pixel_object.contiguous()
    with torch.cuda.device(dev):
        lin = torch.empty((n,), dtype=torch.int32, device=dev)
        sparse = torch.empty((W * H,), dtype=torch.int32, device=dev)
        tiles = torch.empty((((H + 15) // 16) * ((W + 15) // 16),), dtype=torch.int64, device=dev)
        N.check(N.lib().dqo_attach_pixels(n, N.ptr(temp_xyz), N.ptr(viewmatrix), fx, fy, cx, cy, W, H, N.ptr(pixel_object), N.ptr(lin),
                                          N.ptr(sparse), N.ptr(tiles), N.current_stream()))
    return lin, sparse, tiles


def attach_decide(temp_xyz, temp_opacity, temp_obj, lin, hit_index, hit_weight, stable_xyz, scaling_raw, rotation_raw, stable_obj,
                  add_depth_thres, unstable_opacity_low=0.1):
    """dqo_attach_decide (include/dqo_raster.h): uint8 [N], 1 = the candidate attaches — temp_points_attach_mask_per_object in one
    launch (normals from the raw quaternion and raw scales inside)."""
    import _dqo_native as N
    n, dev = int(temp_xyz.shape[0]), temp_xyz.device
    temp_obj, stable_obj = temp_obj.to(torch.int32), stable_obj.to(torch.int32)  # (object ids arrive as int32 or int64)
    f32 = (temp_xyz, temp_opacity, hit_weight, stable_xyz, scaling_raw, rotation_raw)
    i32 = (temp_obj, lin, hit_index, stable_obj)
    N.require_gpu(*f32, *i32)
    if any(a.dtype != torch.float32 for a in f32) or any(a.dtype != torch.int32 for a in i32):
        raise RuntimeError("attach_decide: float32 points / opacities / weights / parameters, int32 ids / pixels / hit indices")
    P = int(stable_xyz.shape[0])
    if (temp_opacity.numel() != n or temp_obj.numel() != n or lin.numel() != n or hit_index.numel() != hit_weight.numel()
            or scaling_raw.numel() != 3 * P or rotation_raw.numel() != 4 * P or stable_obj.numel() != P):
        raise RuntimeError("attach_decide: sizes do not agree")
    c = lambda a: a.contiguous()
    with torch.cuda.device(dev):
        out = torch.empty((n,), dtype=torch.uint8, device=dev)
        N.check(N.lib().dqo_attach_decide(n, N.ptr(c(temp_xyz)), N.ptr(c(temp_opacity)), N.ptr(c(temp_obj)), N.ptr(c(lin)),
                                          N.ptr(c(hit_index)), N.ptr(c(hit_weight)), N.ptr(c(stable_xyz)), N.ptr(c(scaling_raw)),
                                          N.ptr(c(rotation_raw)), N.ptr(c(stable_obj)), 0.5 * add_depth_thres, unstable_opacity_low,
                                          N.ptr(out), N.current_stream()))
    return out


def temp_points_attach_mask_per_object(temp_xyz, temp_opacity, temp_obj, uv, inside, image_width, image_height, hit_index, hit_weight,
                                       stable_xyz, stable_normal, stable_obj, add_depth_thres, unstable_opacity_low=0.1):
    """The per-object job's decision of temp_points_attach_indices (temp_obj / stable_obj given) as ONE boolean mask over the candidates,
    without a host synchronisation: every candidate is judged by itself (its own point, its own pixel, a stable Gaussian of its own
    object), so the boolean-index chain of mapper.py:1393-1430 — five device-to-host round trips — becomes masked arithmetic over all N
    rows.  uv / inside: temp_points_pixels(); hit_index / hit_weight: the op's hit_color / hit_color_weight maps of the gated render
    of the stable cloud ([1, H, W] or flat), only read at the candidates' pixels; a pixel of a never-rendered tile (index 0 with weight
    0, the op's zero fill) counts as no hit.  Returns bool [N]; mask.nonzero() equals temp_points_attach_indices(...) of the same
    inputs (tests/test_gpu_mapgrowth.py)."""
    keep = (temp_opacity > unstable_opacity_low).reshape(-1)
    lin = uv[:, 1].clamp(0, image_height - 1) * image_width + uv[:, 0].clamp(0, image_width - 1)
    s = hit_index.reshape(-1)[lin]
    s = torch.where((s == 0) & (hit_weight.reshape(-1)[lin] == 0), torch.full_like(s, -1), s)
    hit = keep & inside & (s >= 0)
    sidx = s.clamp(min=0).long()
    nrm = stable_normal(sidx) if callable(stable_normal) else stable_normal[sidx]
    d = ((stable_xyz[sidx] - temp_xyz) * nrm).sum(dim=-1)
    return hit & (d.abs() < 0.5 * add_depth_thres) & (stable_obj[sidx] == temp_obj)


def temp_points_attach_indices(temp_xyz, temp_opacity, w2c, intrinsic, image_width, image_height, stable_color_index_map, stable_xyz,
                               stable_normal, add_depth_thres, unstable_opacity_low=0.1, temp_obj=None, stable_obj=None):
    """mapper.py:1384-1430: indices (into the temp cloud) of the points whose opacity the reference sets to `unstable_opacity_low`.

    temp_xyz [N,3], temp_opacity [N,1] (activated); w2c [4,4], intrinsic [3,3] (scene/cameras.py:207-214 get_uv: pixel =
    trunc(K (R x + t) / z), truncation toward zero, so a point up to one pixel left of / above the image still counts as inside);
    stable_color_index_map int [1,H,W] = the op's hit_color of a render of the STABLE Gaussians only (-1 / 0-fill rules as the op has
    them: `>= 0` is the reference's test); stable_xyz [S,3], stable_normal [S,3] (gaussian_pointcloud.py:780-791).

    temp_obj / stable_obj (optional, both): object ids of the temp points and of the map rows; a candidate then only attaches to a stable
    Gaussian of its own object.

    Quirk kept (B15): after the opacity filter the reference indexes the UNFILTERED temp cloud with positions of the FILTERED one when
    it fetches the points for the point-to-plane test (mapper.py:1419); the two agree whenever no temp point has been attached yet
    (all opacities 0.99), which is the state the reference calls it in."""
    n = temp_xyz.shape[0]
    origin = torch.arange(n, device=temp_xyz.device)
    keep = (temp_opacity > unstable_opacity_low).reshape(-1)
    xyz = temp_xyz[keep]
    xyz_c = xyz @ w2c[:3, :3].T + w2c[:3, 3]
    uv = xyz_c @ intrinsic.T
    uv = (uv[:, :2] / uv[:, 2:]).long()
    idx = torch.arange(xyz.shape[0], device=temp_xyz.device)
    inside = (uv[:, 0] >= 0) & (uv[:, 0] < image_width) & (uv[:, 1] >= 0) & (uv[:, 1] < image_height)
    stable_index = stable_color_index_map.permute(1, 2, 0)  # [H,W,1]
    puv = uv[inside]
    hit = stable_index[puv[:, 1], puv[:, 0]] >= 0
    idx = idx[inside][hit[:, 0]]
    sidx = stable_index[uv[idx, 1], uv[idx, 0]].squeeze(-1).long()
    nrm = stable_normal(sidx) if callable(stable_normal) else stable_normal[sidx]  # (a callable: normals of the rows asked for only)
    # (temp_xyz, not xyz: quirk B15 — position idx of the UNFILTERED cloud.  The per-object job takes the candidate's own point instead:
    # which point sits at that position depends on the other candidates of the batch, i.e. on the shard layout)
    d = ((stable_xyz[sidx] - (temp_xyz if temp_obj is None else xyz)[idx]) * nrm).sum(dim=-1)
    on_plane = d.abs() < 0.5 * add_depth_thres
    if temp_obj is not None:
        # the per-object job: a candidate attaches to a stable Gaussian of its OWN object only (temp_obj [N], stable_obj [S]) — on a
        # shard the other objects' Gaussians are not there to attach to
        on_plane &= stable_obj[sidx] == temp_obj[keep][idx]
    idx = idx[on_plane]
    return origin[keep][idx]
