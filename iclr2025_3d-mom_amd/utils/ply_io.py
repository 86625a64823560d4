"""Minimal binary PLY reader/writer for the `vertex` element with float32 properties -- the only PLY the 4DGS
path touches (reference scene/gaussian_model.py:342-407 via plyfile, which is not installed here)."""
import numpy as np


def write_ply(path, names, data):
    data = np.ascontiguousarray(data, dtype="<f4")
    assert data.ndim == 2 and data.shape[1] == len(names)
    hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {data.shape[0]}"]
    hdr += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        f.write(data.tobytes())


def read_ply(path):
    with open(path, "rb") as f:
        names, n, fmt = [], 0, None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property"):
                _, ty, name = line.split()
                assert ty in ("float", "float32"), f"unsupported PLY property type {ty}"
                names.append(name)
            elif line == "end_header":
                break
        assert fmt == "binary_little_endian", fmt
        data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
    return names, data.astype(np.float32)
