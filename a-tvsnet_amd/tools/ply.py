"""Binary PLY point clouds in the layout the reference's fusibile writes (fusibile/displayUtils.h:80-135):
`format binary_little_endian 1.0`, vertices of (float x, y, z, uchar red, green, blue)."""
import numpy as np

_VERTEX = np.dtype([('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('red', 'u1'), ('green', 'u1'), ('blue', 'u1')])


def write_ply(path, points, colors):
    """points (M,3) float32, colors (M,3) uint8 (r, g, b).  Non-finite coordinates are written as (0,0,0) like the
    reference (:115-119)."""
    points = np.asarray(points, np.float32).reshape(-1, 3).copy()
    colors = np.asarray(colors, np.uint8).reshape(-1, 3)
    bad = ~np.isfinite(points).all(axis=1)
    points[bad] = 0.0
    v = np.empty(len(points), _VERTEX)
    v['x'], v['y'], v['z'] = points[:, 0], points[:, 1], points[:, 2]
    v['red'], v['green'], v['blue'] = colors[:, 0], colors[:, 1], colors[:, 2]
    with open(path, 'wb') as f:
        f.write(('ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n'
                 'property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n'
                 % len(points)).encode('ascii'))
        v.tofile(f)


def read_ply(path):
    """-> (points (M,3) float32, colors (M,3) uint8) of a file written by write_ply / the reference."""
    with open(path, 'rb') as f:
        n = None
        while True:
            line = f.readline().decode('ascii').strip()
            if line.startswith('element vertex'):
                n = int(line.split()[-1])
            if line == 'end_header':
                break
        v = np.fromfile(f, _VERTEX, n)
    return np.stack([v['x'], v['y'], v['z']], -1), np.stack([v['red'], v['green'], v['blue']], -1)
