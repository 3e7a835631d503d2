"""CPU restatement of the reference's depth-map fusion (fusibile), float32 NumPy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/fusibile/fusibile.cu:138-277 (the
consistency-voting kernel `fusibile`), :279-325 (copy_point_cloud_to_host), the camera set-up of
/root/reference/fusibile/cameraGeometryUtils.h:194-499 as main.cpp:709-714 calls it (transformP = false, cam_scale 1,
every view selected, main.cpp:721) and the texture set-up of main.cpp:459-498.

PARITY UNPINNED: the reference is CUDA + OpenCV (neither is in this image), it ships no test or golden output, and its
arithmetic is not reproducible bit for bit anyway (nvcc contracts a*b+c into FMAs at its own discretion; the texture
unit interpolates with 8-bit weights).  What this restatement fixes, and what the HIP kernel must reproduce BIT FOR BIT:

* float32 arithmetic, every expression evaluated left to right without contraction;
* tex2D<float4>(tex, x + 0.5, y + 0.5) on an un-normalised, linearly filtered texture = the bilinear blend of texels
  i = floor(x), i + 1 with weight alpha = frac(x) rounded to 8 fractional bits (CUDA programming guide, "Texture
  fetching": 9-bit fixed point with 8 bits of fraction), texel indices clamped to the image (cudaAddressModeWrap only
  acts on normalised coordinates; main.cpp:486-492 asks for it with normalizedCoords = 0, which the runtime treats as
  clamp).  The reference pixel itself is read at alpha = 0: the texel;
* quirks kept: the fused point is the REFERENCE pixel's own 3-D point (the averaging of positions is commented out,
  fusibile.cu:236-237, 258), only normals and colours are averaged; the disparity test is relative
  (|d_proj - d_other| / d_proj < disp_thresh, :221) with d = f_ref * |C_ref - C_i| / depth; `used_pixels` is never
  set; a pixel whose point has a zero coordinate is dropped by the host copy (:309).
"""
import numpy as np

F = np.float32


def decompose_projection(P):
    """K (positive diagonal), R, C of a 3x4 projection matrix (what cv::decomposeProjectionMatrix returns; only
    K[0,0] -- the focal length -- and C are used by the fusion)."""
    P = np.asarray(P, np.float64)
    M = P[:, :3]
    # RQ decomposition M = K R with K upper triangular
    Q, U = np.linalg.qr(np.flipud(M).T)
    K = np.flipud(np.fliplr(U.T))
    R = np.flipud(Q.T)
    S = np.diag(np.sign(np.diag(K)))
    K, R = K @ S, S @ R
    C = -np.linalg.solve(M, P[:, 3])
    return K, R, C


def camera(P):
    """The Camera_cu fields the kernel reads (cameraGeometryUtils.h:377-433): P (12), M_inv (9), C (3), P_col34 (3), f."""
    P = np.asarray(P, np.float32)
    K, _, _ = decompose_projection(P)
    M = P[:, :3].astype(np.float64)
    Minv = np.linalg.inv(M)
    # getCameraCenter (cameraGeometryUtils.h:20-50): signed 3x3 minors of P, then divided by the fourth
    P64 = P.astype(np.float64)
    det = lambda cols: np.linalg.det(P64[:, cols])      # noqa: E731
    C4 = np.array([det([1, 2, 3]), -det([0, 2, 3]), det([0, 1, 3]), -det([0, 1, 2])])
    C = (C4[:3] / C4[3])
    return {'P': P.reshape(12).copy(), 'M_inv': Minv.astype(np.float32).reshape(9), 'C': C.astype(np.float32),
            'P_col34': P[:, 3].copy(), 'f': np.float32(K[0, 0] / K[2, 2])}


def pack_cameras(Ps):
    """(N, 28) float32: P[12] | M_inv[9] | C[3] | P_col34[3] | f -- the layout atvs_fusibile takes."""
    out = np.zeros((len(Ps), 28), np.float32)
    for i, P in enumerate(Ps):
        c = camera(P)
        out[i, 0:12], out[i, 12:21], out[i, 21:24], out[i, 24:27], out[i, 27] = c['P'], c['M_inv'], c['C'], c['P_col34'], c['f']
    return out


def _tex(tex, x, y):
    """tex2D<float4>(tex, x + 0.5, y + 0.5), see the module docstring.  tex (rows, cols, 4); x, y float32 arrays."""
    rows, cols = tex.shape[:2]
    xf, yf = np.floor(x), np.floor(y)
    ax = np.floor((x - xf) * F(256.0) + F(0.5)) / F(256.0)
    ay = np.floor((y - yf) * F(256.0) + F(0.5)) / F(256.0)
    x0 = np.clip(xf.astype(np.int64), 0, cols - 1)
    y0 = np.clip(yf.astype(np.int64), 0, rows - 1)
    x1 = np.clip(xf.astype(np.int64) + 1, 0, cols - 1)
    y1 = np.clip(yf.astype(np.int64) + 1, 0, rows - 1)
    ax, ay = ax[..., None].astype(F), ay[..., None].astype(F)
    one = F(1.0)
    top = (one - ax) * tex[y0, x0] + ax * tex[y0, x1]
    bot = (one - ax) * tex[y1, x0] + ax * tex[y1, x1]
    return ((one - ay) * top + ay * bot).astype(F)


def fuse_reference(cams, normals_depths, images, ref, disp_thresh, normal_thresh, num_consistent):
    """The kernel for one reference camera: -> (coord (rows,cols,3), normal (rows,cols,3), texture (rows,cols,4),
    created (rows,cols) bool).  cams (N,28); normals_depths / images (N,rows,cols,4) float32."""
    N, rows, cols = normals_depths.shape[:3]
    with np.errstate(all='ignore'):
        cr = cams[ref]
        Pr, Mi, Cr, pc = cr[0:12], cr[12:21], cr[21:24], cr[24:27]
        f = cr[27]
        ys, xs = np.meshgrid(np.arange(rows, dtype=F), np.arange(cols, dtype=F), indexing='ij')
        nd = normals_depths[ref]
        normal, depth = nd[..., :3], nd[..., 3]
        # get3Dpoint_cu (:53-62)
        ptx, pty, ptz = depth * xs - pc[0], depth * ys - pc[1], depth - pc[2]
        X = np.stack([Mi[0] * ptx + Mi[1] * pty + Mi[2] * ptz,
                      Mi[3] * ptx + Mi[4] * pty + Mi[5] * ptz,
                      Mi[6] * ptx + Mi[7] * pty + Mi[8] * ptz], -1).astype(F)
        cons_n = np.concatenate([normal, nd[..., 3:4]], -1).copy()      # float4 normal (w rides along, :164)
        cons_t = images[ref].copy()
        count = np.zeros((rows, cols), np.int32)
        for i in range(N):
            if i == ref:
                continue
            c = cams[i]
            P = c[0:12]
            # project_on_camera (:127-133)
            tx = P[0] * X[..., 0] + P[1] * X[..., 1] + P[2] * X[..., 2] + P[3]
            ty = P[4] * X[..., 0] + P[5] * X[..., 1] + P[6] * X[..., 2] + P[7]
            tz = P[8] * X[..., 0] + P[9] * X[..., 1] + P[10] * X[..., 2] + P[11]
            px, py, d = (tx / tz).astype(F), (ty / tz).astype(F), tz.astype(F)
            inb = (px >= 0) & (px < cols) & (py >= 0) & (py < rows)
            pxs, pys = np.where(inb, px, F(0)), np.where(inb, py, F(0))
            other = _tex(normals_depths[i], pxs, pys)
            dC = Cr - c[21:24]
            base = np.sqrt(dC[0] * dC[0] + dC[1] * dC[1] + dC[2] * dC[2]).astype(F)
            fb = f * base
            d_disp = (fb / d).astype(F)
            o_disp = (fb / other[..., 3]).astype(F)
            ok = inb & ((np.abs(d_disp - o_disp) / d_disp) < F(disp_thresh))
            dot = other[..., 0] * normal[..., 0] + other[..., 1] * normal[..., 1] + other[..., 2] * normal[..., 2]
            ang = np.arccos(dot.astype(F)).astype(F)
            ang = np.where(ang != ang, F(0), ang)
            ok &= ang < F(normal_thresh)
            add4 = lambda a, b: np.concatenate([a[..., :3] + b[..., :3], np.zeros_like(a[..., :1])], -1)   # noqa: E731
            cons_n = np.where(ok[..., None], add4(cons_n, other), cons_n)
            cons_t = np.where(ok[..., None], add4(cons_t, _tex(images[i], pxs, pys)), cons_t)
            count += ok
        k = count.astype(F) + F(1.0)
        cons_n = (cons_n[..., :3] / k[..., None]).astype(F)
        cons_t = np.concatenate([cons_t[..., :3] / k[..., None], np.zeros_like(cons_t[..., :1])], -1).astype(F)
        created = count >= int(num_consistent)
    return X, cons_n, cons_t, created


def fuse(Ps, depths, normals, images, disp_thresh=0.01, normal_thresh=2.0 * np.pi, num_consistent=2):
    """runcuda + copy_point_cloud_to_host over every camera (fusibile.cu:422-427, 279-325).

    Ps: N 3x4 projection matrices; depths (N,rows,cols); normals (N,rows,cols,3); images (N,rows,cols,3) BGR (uint8 or
    float).  -> (points (M,3) float32, colors (M,3) uint8 RGB), in camera-major, row-major order like the reference."""
    cams = pack_cameras(Ps)
    N = len(Ps)
    nd = np.concatenate([np.asarray(normals, F), np.asarray(depths, F)[..., None]], -1)
    img4 = np.concatenate([np.asarray(images).astype(F), np.zeros(np.asarray(images).shape[:3] + (1,), F)], -1)
    pts, cols_ = [], []
    for ref in range(N):
        X, _, tex, created = fuse_reference(cams, nd, img4, ref, disp_thresh, normal_thresh, num_consistent)
        keep = created & (X[..., 0] != 0) & (X[..., 1] != 0) & (X[..., 2] != 0)
        pts.append(X[keep])
        t = tex[keep]
        # storePlyFileBinaryPointCloud (displayUtils.h:109-111): (char)(int) of channels 2, 1, 0
        cols_.append(np.stack([t[:, 2], t[:, 1], t[:, 0]], -1).astype(np.int32).astype(np.uint8))
    return np.concatenate(pts, 0), np.concatenate(cols_, 0)
