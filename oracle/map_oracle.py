"""ORACLE (test infrastructure only) for the fused mapping-step helpers: numpy restatement of

  * the masked loss of /root/reference/SLAM/multiprocess/mapper.py:836-875 (render_mask given => no SSIM term, B14;
    normal_weight = 0) and its gradient w.r.t. the rendered colour / depth,
  * the activations of SLAM/gaussian_pointcloud.py:732-733, 746-747 and their Jacobians,
  * one torch.optim.Adam step (mapper.py:548: Adam(lr=0 default, eps=1e-15); groups gaussian_pointcloud.py:338-370).

PARITY STATUS: pinned — the reference's own implementation of these pieces IS torch (eager ops + torch.optim.Adam), which runs
on CPU here: tests/test_oracle_map.py checks this restatement against torch autograd + torch.optim.Adam on the same inputs.
"""
import numpy as np


def masked_loss(color, depth, depth_index, gt_color, gt_depth, render_mask, color_weight=0.8, depth_weight=1.0, add_depth_thres=0.1):
    """Returns (total, color_loss, depth_loss, dL_dcolor[3,H,W], dL_ddepth[1,H,W]) in float64."""
    color, depth, gt_color, gt_depth = (np.asarray(a, np.float64) for a in (color, depth, gt_color, gt_depth))
    m = np.ones(depth.shape[1:], bool) if render_mask is None else np.asarray(render_mask).astype(bool)
    n_col = max(int(m.sum()), 1)
    d = color - gt_color
    color_loss = np.abs(d)[:, m].sum() / (3.0 * n_col)                       # l1_loss(image[mask], gt[mask]) (mapper.py:847)
    err = depth - gt_depth
    valid = (np.asarray(depth_index) != -1) & (gt_depth > 0) & (err < add_depth_thres) & m[None]   # mapper.py:850-856
    n_dep = max(int(valid.sum()), 1)
    depth_loss = np.abs(err[valid]).sum() / n_dep
    total = depth_weight * depth_loss + color_weight * color_loss
    dL_dcolor = np.sign(d) * m[None] * (color_weight / (3.0 * n_col))
    dL_ddepth = np.sign(err) * valid * (depth_weight / n_dep)
    return total, color_loss, depth_loss, dL_dcolor, dL_ddepth


def per_object_masked_loss(color, depth, depth_index, gt_color, gt_depth, pixel_object, render_mask=None, color_weight=0.8,
                           depth_weight=1.0, add_depth_thres=0.1):
    """The loss of the per-object job (DqoLossTap.per_object):  sum over the object ids k of masked_loss evaluated on
    object k's pixels alone (pixel_object == k inside render_mask), each term with its own pixel counts.  Returns
    (total, color_loss, depth_loss, dL_dcolor[3,H,W], dL_ddepth[1,H,W]) in float64."""
    po = np.asarray(pixel_object).reshape(np.asarray(depth).shape[1:])
    m = po >= 0
    if render_mask is not None:
        m = m & np.asarray(render_mask).astype(bool)
    tot = col = dep = 0.0
    dC, dD = np.zeros(np.asarray(color).shape, np.float64), np.zeros(np.asarray(depth).shape, np.float64)
    for k in np.unique(po[m]):
        t, c, d, gc, gd = masked_loss(color, depth, depth_index, gt_color, gt_depth, m & (po == k), color_weight, depth_weight, add_depth_thres)
        tot, col, dep = tot + t, col + c, dep + d
        dC += gc
        dD += gd
    return tot, col, dep, dC, dD


def activate(opacity_raw, scaling_raw, rotation_raw):
    q = np.asarray(rotation_raw, np.float64)
    n = np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    return 1.0 / (1.0 + np.exp(-np.asarray(opacity_raw, np.float64))), np.exp(np.asarray(scaling_raw, np.float64)), q / n


def raw_grads(opacity_raw, scaling_raw, rotation_raw, g_opacity, g_scales, g_rot):
    """Chain rule through sigmoid / exp / F.normalize."""
    sg = 1.0 / (1.0 + np.exp(-np.asarray(opacity_raw, np.float64)))
    q = np.asarray(rotation_raw, np.float64)
    n = np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    y = q / n
    g = np.asarray(g_rot, np.float64)
    return (np.asarray(g_opacity, np.float64) * sg * (1 - sg), np.asarray(g_scales, np.float64) * np.exp(np.asarray(scaling_raw, np.float64)),
            (g - y * (y * g).sum(1, keepdims=True)) / n)


def attach_loss(scaling, xyz, rotation, scaling0, xyz0, rotation0, opacity_raw0):
    """Attach loss of Mapping.loss_update, /root/reference/SLAM/multiprocess/mapper.py:812-829, with l2_loss = mean of squares
    (utils/loss_utils.py:34-36):  a = sigmoid(opacity at the start of the call) < 0.9;
        1000 * (mean((scaling[a] - scaling0[a])^2) + mean((xyz[a] - xyz0[a])^2) + mean((rotation[a] - rotation0[a])^2)),
    0 when a is empty.  Returns (loss, d/dscaling, d/dxyz, d/drotation) on the RAW parameters, float64."""
    f = lambda x: np.asarray(x, np.float64)
    a = (1.0 / (1.0 + np.exp(-f(opacity_raw0).reshape(-1)))) < 0.9
    n = int(a.sum())
    outs, loss = [], 0.0
    for p, p0 in ((scaling, scaling0), (xyz, xyz0), (rotation, rotation0)):
        d = (f(p) - f(p0)) * a[:, None]
        if n == 0:
            outs.append(np.zeros_like(d))
            continue
        loss += 1000.0 * (d ** 2).sum() / (n * d.shape[1])
        outs.append(2000.0 * d / (n * d.shape[1]))
    return (loss, *outs)


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-15):
    m = m + (g - m) * (1 - beta1)
    v = v * beta2 + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p = p - (lr / bc1) * (m / (np.sqrt(v) / np.sqrt(bc2) + eps))
    return p, m, v


def accumulate_gaussian_error(H, W, P, color_err, depth_err, normal_err, color_index, depth_index, color_thr, depth_thr, normal_thr,
                              check_max=True):
    """/root/reference/submodules/cuda_utils/map_process.cu:33-110 (+ :146-172 mean) restated with numpy scatter ops.
    PARITY STATUS: unpinned by reference tests (CUDA source, no fixtures); max / count results are exact integer-like
    operations (order-independent), the mean mode is compared with a tolerance."""
    f = np.float32
    ce, de, ne = (np.asarray(a, f).reshape(-1)[:H * W] for a in (color_err, depth_err, normal_err))
    ci, di = (np.asarray(a, np.int64).reshape(-1)[:H * W] for a in (color_index, depth_index))
    gc, gd, gn, rs = (np.zeros(P, f) for _ in range(4))
    cm, dm = (ci >= 0) & (ci < P), (di >= 0) & (di < P)
    if check_max:
        # float atomicMax against a 0 initial value: NaN never wins (val > old is false)
        np.maximum.at(gc, ci[cm & ~np.isnan(ce)], ce[cm & ~np.isnan(ce)])
        np.maximum.at(gd, di[dm & ~np.isnan(de)], de[dm & ~np.isnan(de)])
        np.maximum.at(gn, di[dm & ~np.isnan(ne)], ne[dm & ~np.isnan(ne)])
    else:
        s = [np.zeros(P, np.float64) for _ in range(3)]
        np.add.at(s[0], ci[cm], ce[cm])
        np.add.at(s[1], di[dm], de[dm])
        np.add.at(s[2], di[dm], ne[dm])
        cc, dc = np.bincount(ci[cm], minlength=P), np.bincount(di[dm], minlength=P)
        gc = np.where(cc > 0, s[0] / np.maximum(cc, 1), s[0]).astype(f)
        gd = np.where(dc > 0, s[1] / np.maximum(dc, 1), s[1]).astype(f)
        gn = np.where(dc > 0, s[2] / np.maximum(dc, 1), s[2]).astype(f)
    np.add.at(rs, ci[cm & (ce > color_thr)], 1)
    np.add.at(rs, di[dm & (de > depth_thr)], 1)
    np.add.at(rs, di[dm & (ne > normal_thr)], 1)
    return gc.reshape(P, 1), gd.reshape(P, 1), gn.reshape(P, 1), rs.reshape(P, 1)


def accumulate_gaussian_confidence(H, W, P, index_map, confidence_map):
    """/root/reference/submodules/cuda_utils/map_process.cu:247-360 + cuda_utils.cu:62-83 restated with numpy scatter ops:
    (max [P,1], min [P,1], mean [P,1]) of the confidence over the pixels that name each Gaussian; 0 / 0 / 0 where none does.  The
    reference's float atomicMax / atomicMin replace on a strict comparison only, so NaN never wins.  PARITY STATUS: unpinned by
    reference tests (CUDA source, no fixtures, no caller); max / min are order-independent and exact, the mean is compared with a
    tolerance."""
    f = np.float32
    conf = np.asarray(confidence_map, f).reshape(-1)[:H * W]
    idx = np.asarray(index_map, np.int64).reshape(-1)[:H * W]
    ok = (idx >= 0) & (idx < P)
    gmax = np.full(P, np.finfo(f).min, f)
    gmin = np.full(P, np.finfo(f).max, f)
    fin = ok & ~np.isnan(conf)
    np.maximum.at(gmax, idx[fin], conf[fin])
    np.minimum.at(gmin, idx[fin], conf[fin])
    s = np.zeros(P, np.float64)
    np.add.at(s, idx[ok], conf[ok].astype(np.float64))
    n = np.bincount(idx[ok], minlength=P)
    seen = n > 0
    mean = np.where(seen, s / np.maximum(n, 1), 0.0).astype(f)
    gmax, gmin = np.where(seen, gmax, f(0)), np.where(seen, gmin, f(0))
    return gmax.reshape(P, 1), gmin.reshape(P, 1), mean.reshape(P, 1)


def temp_points_attach_indices(temp_xyz, temp_opacity, w2c, intrinsic, W, H, stable_color_index_map, stable_xyz, stable_normal,
                               add_depth_thres, low=0.1):
    """/root/reference/SLAM/multiprocess/mapper.py:1384-1430 + scene/cameras.py:207-214 restated as a per-point loop (small cases)."""
    out = []
    f = np.float32
    kept = [i for i in range(len(temp_xyz)) if temp_opacity[i, 0] > f(low)]
    for pos, i in enumerate(kept):
        pc = (w2c[:3, :3].astype(f) @ temp_xyz[i].astype(f) + w2c[:3, 3].astype(f)).astype(f)
        q = (intrinsic.astype(f) @ pc).astype(f)
        u, v = int(np.trunc(f(q[0]) / f(q[2]))), int(np.trunc(f(q[1]) / f(q[2])))
        if not (0 <= u < W and 0 <= v < H):
            continue
        s = int(stable_color_index_map[0, v, u])
        if s < 0:
            continue
        p = temp_xyz[pos]  # the reference reads the UNFILTERED cloud at the FILTERED position (mapper.py:1419)
        d = float(np.sum(((stable_xyz[s] - p) * stable_normal[s]).astype(f), dtype=f))
        if abs(d) < 0.5 * add_depth_thres:
            out.append(i)
    return np.array(out, np.int64)


# ---------------------------------------------------------------- row f3: tile-mask producers ---------------------------
def _pad_tiles(img, stride, value=0):
    """F.pad(x, (0, pad_w, 0, pad_h), value) of SLAM/utils.py:721-724 and friends."""
    h, w = img.shape[:2]
    H, W = (h + stride - 1) // stride * stride, (w + stride - 1) // stride * stride
    out = np.full((H, W), value, dtype=img.dtype)
    out[:h, :w] = img
    return out


def pixelmask2tilemask(pixelmask, stride):
    """SLAM/utils.py:731-743: max-pool of the zero-padded mask -> int32 [gy, gx]."""
    p = _pad_tiles((np.asarray(pixelmask) != 0).astype(np.float32), stride)
    gy, gx = p.shape[0] // stride, p.shape[1] // stride
    return p.reshape(gy, stride, gx, stride).max((1, 3)).astype(np.int32)


def meanpool(matrix, stride):
    """SLAM/utils.py:720-729: avg_pool2d of the zero-padded image (count_include_pad: always / stride^2), float32."""
    p = _pad_tiles(np.asarray(matrix, np.float32), stride)
    gy, gx = p.shape[0] // stride, p.shape[1] // stride
    return (p.reshape(gy, stride, gx, stride).astype(np.float64).sum((1, 3)) / (stride * stride)).astype(np.float32)


def transmission2tilemask(pixelmask, stride, tile_mask_ratio=0.5):
    """SLAM/utils.py:752-763."""
    return (meanpool((np.asarray(pixelmask) != 0).astype(np.float32), stride) > np.float32(tile_mask_ratio)).astype(np.int32)


def color_error_image(render, gt):
    """mapper.py:949-956: sum over channels of |render - gt|, zero where the rendered colour sums to 0 (float32, torch order)."""
    r, g = np.asarray(render, np.float32), np.asarray(gt, np.float32)
    d = np.abs(r - g)
    e = (d[0] + d[1]) + d[2]
    e[((r[0] + r[1]) + r[2]) == 0] = 0
    return e


def colorerror2tilemask(color_error, stride, top_ratio=0.4):
    """SLAM/utils.py:766-799.  Returns (mask int32 [gy, gx], pooled float32, k): ties at the k-th value make the reference's
    torch.topk choice implementation-defined, so callers compare masks only away from that value."""
    pooled = meanpool(color_error, stride)
    k = int(pooled.size * top_ratio)
    order = np.argsort(-pooled.reshape(-1), kind="stable")
    mask = np.zeros(pooled.size, np.int32)
    mask[order[:k]] = 1
    return mask.reshape(pooled.shape), pooled, k


# ---------------------------------------------------------------- row f3: 3-NN with query != reference set -----------------
def knn3_query(q, r, block=512, brute_limit=5e7):
    """pytorch3d.ops.knn_points(q[None], r[None], K=3, norm=2) restated: exact squared L2 distances (float32, (dx^2 + dy^2) + dz^2
    like the kernel) of the 3 nearest references, ascending, and their indices (ties: lower index first).  Brute force up to
    brute_limit pairs; beyond that the 12 nearest candidates come from scipy's cKDTree (fp64) and are re-ranked with the same
    fp32 arithmetic (fp32 rounding cannot promote a candidate from beyond the 12th place)."""
    q, r = np.asarray(q, np.float32), np.asarray(r, np.float32)
    Q, R = len(q), len(r)
    k = min(3, R)
    dist = np.full((Q, 3), np.finfo(np.float32).max, np.float32)
    idx = np.full((Q, 3), -1, np.int64)

    def d2_of(qq, cand):  # qq [n, 3], cand [n, m, 3]
        d = qq[:, None, :] - cand
        d = d * d
        return (d[..., 0] + d[..., 1]) + d[..., 2]

    if Q * R <= brute_limit or R <= 12:
        for s in range(0, Q, block):
            d2 = d2_of(q[s:s + block], r[None, :, :])
            order = np.argsort(d2, axis=1, kind="stable")[:, :k]
            idx[s:s + block, :k] = order
            dist[s:s + block, :k] = np.take_along_axis(d2, order, 1)
        return dist, idx
    from scipy.spatial import cKDTree
    _, cand = cKDTree(r.astype(np.float64)).query(q.astype(np.float64), k=12)
    cand = np.sort(cand, axis=1)  # ascending index so that the stable sort breaks distance ties by lower index
    d2 = d2_of(q, r[cand])
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]
    idx[:, :k] = np.take_along_axis(cand, order, 1)
    dist[:, :k] = np.take_along_axis(d2, order, 1)
    return dist, idx


# ---------------------------------------------------------------- row f4: ICP normal equations ------------------------------
def icp_normal_equations(vertex0, vertex1, normal0, normal1, pose10, K, distance_threshold, normal_threshold):
    """SLAM/icp.py:51-123 restated: per-pixel fp32 like the reference, sums in fp64.  Returns (JtJ [6,6], JtR [6], valid mask [H,W])."""
    f = np.float32
    v0, v1, n0, n1 = (np.asarray(a, f) for a in (vertex0, vertex1, normal0, normal1))
    pose = np.asarray(pose10, f)
    R, t = pose[:3, :3], pose[:3, 3]
    H, W, _ = v0.shape
    p = (v0.reshape(-1, 3) @ R.T).astype(f).reshape(H, W, 3) + t[None, None, :]
    n = (n0.reshape(-1, 3) @ R.T).astype(f).reshape(H, W, 3)
    fx, fy, cx, cy = f(K[0][0]), f(K[1][1]), f(K[0][2]), f(K[1][2])
    with np.errstate(divide="ignore", invalid="ignore"):
        u = (p[..., 0] / p[..., 2]) * fx + cx
        v = (p[..., 1] / p[..., 2]) * fy + cy
        inview = (u > 0) & (u < W - 1) & (v > 0) & (v < H - 1)
        # grid_sample(mode="nearest", padding_mode="border", align_corners=True) of warp_features (icp.py:131-149)
        un, vn = u / f((W - 1) / 2) - f(1), v / f((H - 1) / 2) - f(1)
        gx = np.clip(((un + f(1)) / f(2)) * f(W - 1), 0, W - 1)
        gy = np.clip(((vn + f(1)) / f(2)) * f(H - 1), 0, H - 1)
        xi = np.nan_to_num(np.rint(gx), nan=0.0).astype(np.int64).clip(0, W - 1)
        yi = np.nan_to_num(np.rint(gy), nan=0.0).astype(np.int64).clip(0, H - 1)
    q, m = v1[yi, xi], n1[yi, xi]
    diff = p - q
    ndm = (n * m).sum(-1) > f(normal_threshold)
    res = (m * diff).sum(-1)
    J = np.concatenate([np.cross(p, m), m], -1)  # J_rot = -(m^T [p]_x) = p x m
    with np.errstate(invalid="ignore"):
        occ = ~inview | (np.sqrt((diff * diff).sum(-1)) > f(distance_threshold))
    valid = ~(occ | ~(v0[..., 2] > 0) | ~(q[..., 2] > 0) | ~ndm)
    Jv = J[valid].astype(np.float64)
    rv = res[valid].astype(np.float64)
    return Jv.T @ Jv, Jv.T @ rv, valid


def torch_lerp(a, b, w):
    """aten lerp (torch.lerp): weight < 0.5 ? a + w (b - a) : b - (b - a) (1 - w), elementwise, in the arrays' dtype."""
    d = b - a
    return np.where(np.abs(w) < 0.5, a + w * d, b - d * (1 - w))


def slerp(v0, v1, t, dot_threshold=0.9995):
    """/root/reference/SLAM/utils.py:650-709 restated (v0, v1 [P, 4]; t [P, 1]): rows whose normalised dot product is NaN or beyond the
    threshold in magnitude are torch.lerp'ed, the others sin-weighted — no shortest-arc flip, no renormalisation of the result."""
    n0 = np.sqrt((v0 * v0).sum(-1, keepdims=True))
    n1 = np.sqrt((v1 * v1).sum(-1, keepdims=True))
    with np.errstate(invalid="ignore", divide="ignore"):
        dot = ((v0 / n0) * (v1 / n1)).sum(-1)
        lerp_rows = np.isnan(dot) | (np.abs(dot) > dot_threshold)
        th0 = np.arccos(dot)[:, None]
        s_th0 = np.sin(th0)
        tht = th0 * t
        s0, s1 = np.sin(th0 - tht) / s_th0, np.sin(tht) / s_th0
        out = np.where(lerp_rows[:, None], torch_lerp(v0, v1, t), s0 * v0 + s1 * v1)
    return out.astype(v0.dtype)


def history_merge(hist, cur, max_weight=0.5, dtype=np.float32):
    """Mapping.history_merge, /root/reference/SLAM/multiprocess/mapper.py:607-650, on ONE cloud (the trained one).  hist / cur: dicts with
    confidence [P,1], xyz [P,3], features_dc [P,1,3], features_rest [P,M-1,3], scaling [P,3]; hist["rotation"] = the ACTIVATED rotation at
    the start of the call, cur["rotation_raw"] the raw one now.  Returns dict(xyz, features_dc, features_rest, scaling, rotation) — the
    new raw parameters.  Reproduces the reference's `history_weight[0]` quirk (:620-637): features and scaling of EVERY row are merged
    with the FIRST row's weight.  PARITY: the slerp inside is pinned against the imported reference function
    (tests/golden/make_history_merge_golden.py); the surrounding lerps are the ten statements restated here."""
    f = lambda a: np.asarray(a, dtype)
    if max_weight <= 0:
        return dict(xyz=f(cur["xyz"]), features_dc=f(cur["features_dc"]), features_rest=f(cur["features_rest"]), scaling=f(cur["scaling"]),
                    rotation=f(cur["rotation_raw"]))
    w = dtype(max_weight) * f(hist["confidence"]) / (f(cur["confidence"]) + dtype(1e-6))          # [P,1]
    w0 = w[0]                                                                                      # `history_weight[0]`: shape [1]
    q = f(cur["rotation_raw"])
    rot_now = q / np.maximum(np.sqrt((q * q).sum(-1, keepdims=True)), dtype(1e-12))               # get_rotation = F.normalize
    return dict(xyz=f(hist["xyz"]) * w + (1 - w) * f(cur["xyz"]),
                features_dc=f(hist["features_dc"]) * w0 + (1 - w0) * f(cur["features_dc"]),
                features_rest=f(hist["features_rest"]) * w0 + (1 - w0) * f(cur["features_rest"]),
                scaling=f(hist["scaling"]) * w0 + (1 - w0) * f(cur["scaling"]),
                rotation=slerp(f(hist["rotation"]), rot_now, 1 - w))
