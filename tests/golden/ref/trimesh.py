"""Stand-in for the two trimesh calls the reference's tests make (`trimesh.load(path).vertices` on a binary
little-endian PLY point cloud, `trimesh.PointCloud(vertices=...).export(path)`); trimesh is not installed in this image."""
import numpy as np


class _Cloud:
    def __init__(self, vertices):
        self.vertices = vertices


class PointCloud:
    def __init__(self, vertices=None, colors=None):
        self.vertices = np.asarray(vertices, dtype=np.float64)

    def export(self, path):
        v = self.vertices.reshape(-1, 3)
        with open(path, "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(v))
            np.savetxt(f, v, fmt="%.7g")


def load(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        n, props, fmt = 0, [], None
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property") and "list" not in line:
                props.append(line.split()[1:3])
            elif line == "end_header":
                break
        assert fmt == "binary_little_endian", fmt
        kinds = {"float": "<f4", "float32": "<f4", "double": "<f8", "uchar": "u1", "uint8": "u1", "int": "<i4", "uint": "<u4"}
        dt = np.dtype([(name, kinds[t]) for t, name in props])
        data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    return _Cloud(np.stack([data["x"], data["y"], data["z"]], -1).astype(np.float64))
