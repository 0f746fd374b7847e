"""PLY wire format of DQO-MAP's Gaussian maps — host-side interchange, no GPU work.

Reads and writes the files of /root/reference/SLAM/gaussian_pointcloud.py:
    construct_list_of_attributes :557-588   x y z nx ny nz f_dc_0..2 f_rest_0..(3(D+1)^2-4) opacity scale_0..2 rot_0..3 [confidence]
    save_model_ply :641-684                  one "vertex" element, every property float32, normals written as zeros,
                                             f_dc / f_rest flattened CHANNEL-major ([P, K, 3] -> transpose -> [P, 3 K])
    load :132-207                            the inverse; a missing confidence column reads as zeros
The reference goes through the third-party `plyfile` package (not vendored, not installed here), which writes
`format binary_little_endian 1.0` for native-endian float32 records; this module writes / parses that layout directly with numpy
(and also reads `format ascii 1.0`).  Values are RAW parameters (logit opacity, log scales, unnormalised quaternions), exactly what
the reference stores.
"""
import numpy as np


def attribute_names(n_rest, include_confidence=True):
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    if include_confidence:
        names.append("confidence")
    return names


def save_model_ply(path, xyz, shs, opacity_raw, scaling_raw, rotation_raw, confidence=None, include_confidence=True):
    """xyz [P,3], shs [P,M,3] (coefficient 0 = f_dc, 1.. = f_rest), opacity_raw [P,1], scaling_raw [P,3], rotation_raw [P,4],
    confidence [P,1] or None (zeros).  Tensors or arrays; nothing is written for an empty map (gaussian_pointcloud.py:642-643)."""
    a = lambda t: np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, np.float32)
    xyz, shs, op, sc, rot = a(xyz), a(shs), a(opacity_raw).reshape(-1, 1), a(scaling_raw), a(rotation_raw)
    P = xyz.shape[0]
    if P == 0:
        return
    f_dc = shs[:, :1, :].transpose(0, 2, 1).reshape(P, -1)     # [P,1,3] -> [P,3,1] -> [P,3]
    f_rest = shs[:, 1:, :].transpose(0, 2, 1).reshape(P, -1)   # [P,K,3] -> [P,3,K] -> [P,3K]: channel-major
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, op, sc, rot]
    if include_confidence:
        cols.append(np.zeros((P, 1), np.float32) if confidence is None else a(confidence).reshape(-1, 1))
    table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = attribute_names(f_rest.shape[1], include_confidence)
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {P}\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(table.tobytes())


def _read_table(path):
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is not None:
                    raise ValueError(f"{path}: elements after 'vertex' are not supported")
            elif tok[0] == "property" and in_vertex:
                if tok[1] not in ("float", "float32"):
                    raise ValueError(f"{path}: property {tok[-1]} has type {tok[1]}, the map format is all float32")
                props.append(tok[2])
            elif tok[0] == "end_header":
                break
        if count is None:
            raise ValueError(f"{path}: no vertex element")
        if fmt in ("binary_little_endian", "binary_big_endian"):
            dt = "<f4" if fmt == "binary_little_endian" else ">f4"
            data = np.frombuffer(fh.read(4 * count * len(props)), dtype=dt)
            if data.size != count * len(props):
                raise ValueError(f"{path}: truncated vertex data")
            table = data.reshape(count, len(props)).astype(np.float32)
        elif fmt == "ascii":
            table = np.loadtxt(fh, dtype=np.float32, max_rows=count, ndmin=2)
            if table.shape != (count, len(props)):
                raise ValueError(f"{path}: bad ascii vertex table")
        else:
            raise ValueError(f"{path}: unknown format {fmt}")
    return props, table


def load_model_ply(path, max_sh_degree=3):
    """dict(xyz [P,3], shs [P,(D+1)^2,3], opacity_raw [P,1], scaling_raw [P,3], rotation_raw [P,4], confidence [P,1]) as float32
    arrays — the reference's `load` (gaussian_pointcloud.py:132-207) with f_dc / f_rest merged into one SH tensor."""
    props, table = _read_table(path)
    col = {n: i for i, n in enumerate(props)}
    get = lambda names: table[:, [col[n] for n in names]]
    rest = sorted((n for n in props if n.startswith("f_rest_")), key=lambda n: int(n.split("_")[-1]))
    K = (max_sh_degree + 1) ** 2 - 1
    assert len(rest) == 3 * K, f"{path}: {len(rest)} f_rest columns, degree {max_sh_degree} needs {3 * K}"
    P = table.shape[0]
    f_dc = get(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(P, 3, 1).transpose(0, 2, 1)
    f_rest = get(rest).reshape(P, 3, K).transpose(0, 2, 1)
    scales = sorted((n for n in props if n.startswith("scale_")), key=lambda n: int(n.split("_")[-1]))
    rots = sorted((n for n in props if n.startswith("rot")), key=lambda n: int(n.split("_")[-1]))
    conf = get(["confidence"]) if "confidence" in col else np.zeros((P, 1), np.float32)
    return dict(xyz=get(["x", "y", "z"]), shs=np.ascontiguousarray(np.concatenate([f_dc, f_rest], 1)), opacity_raw=get(["opacity"]),
                scaling_raw=get(scales), rotation_raw=get(rots), confidence=conf)
