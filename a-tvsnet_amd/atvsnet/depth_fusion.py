#!/usr/bin/env python
"""Depth-map filtering and fusion into a point cloud, mirroring the reference's ``atvsnet/depth_fusion.py`` (CLI, file
layout, function names) with the external ``fusibile`` executable replaced by the HIP kernel behind
``atvs_fusibile`` (include/atvsnet_hip.h; reference fusibile/fusibile.cu:138-277, host side fusibile/main.cpp:559-845).

    python -m atvsnet_amd.atvsnet.depth_fusion --dense_folder <scene> [--prob_threshold 0.8]
        [--disp_threshold 0.01] [--num_consistent 2]

reads  <dense_folder>/depths_atvsnet/{%08d.pfm, %08d_prob.pfm, %08d.jpg, %08d.txt}   (written by eval_pointcloud)
writes <dense_folder>/depths_atvsnet/%08d_prob_filtered.pfm                          (probability_filter, :185-206)
       <dense_folder>/points_atvsnet/{cams/*.P, images/*.jpg, 2333__%08d/{disp,normals}.dmb}   (atvsnet_to_gipuma, :115-182)
       <dense_folder>/points_atvsnet/consistencyCheck-<date>-<time>/final3d_model.ply          (fusibile main.cpp:600-604, 838-842)
       <dense_folder>/final3d_model.ply                                              (:262-267)

The gipuma-format intermediate files are kept (same bytes as the reference writes) so that either fusion back-end can
consume them; ``--fusibile_exe_path`` is accepted and ignored.
"""
from __future__ import print_function

import argparse
import os
import shutil
import time

import numpy as np

from .preprocess import load_cam, load_pfm, write_pfm
from ..tools import ply


_DMB_HEADER = np.dtype([('type', '<i4'), ('height', '<i4'), ('width', '<i4'), ('channels', '<i4')])


def read_gipuma_dmb(path):
    """Gipuma .dmb image -> (h, w) or (h, w, c) float32 (reference :24-35).  File = 16-byte header
    (int32 type, height, width, channels) + float32 channel PLANES, each plane row-major (h, w)."""
    raw = np.fromfile(path, dtype=np.uint8)
    head = raw[:_DMB_HEADER.itemsize].view(_DMB_HEADER)[0]
    h, w, c = int(head['height']), int(head['width']), int(head['channels'])
    planes = raw[_DMB_HEADER.itemsize:].view('<f4').reshape(c, h, w)
    return np.moveaxis(planes, 0, -1).squeeze()


def write_gipuma_dmb(path, image):
    """(h, w) or (h, w, c) -> Gipuma .dmb (reference :37-58): header with type = 1, then the channel planes."""
    image = np.asarray(image, np.float32)
    planes = image[None] if image.ndim == 2 else np.moveaxis(image, -1, 0)
    head = np.zeros(1, _DMB_HEADER)
    head['type'], head['height'], head['width'], head['channels'] = (1,) + planes.shape[1:] + planes.shape[:1]
    with open(path, 'wb') as fid:
        fid.write(head.tobytes())
        fid.write(np.ascontiguousarray(planes, '<f4').tobytes())


def atvsnet_to_gipuma_dmb(in_path, out_path):
    '''convert atvsnet .pfm output to Gipuma .dmb format (reference :60-68; also saves a viridis .png)'''
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as matplt
    with open(in_path, 'rb') as f:
        image = load_pfm(f)
    matplt.imsave(out_path[0:-4] + '.png', np.squeeze(image), cmap='viridis')
    write_gipuma_dmb(out_path, image)


def projection_matrix(cam):
    '''P = (K with its fourth row zeroed) @ extrinsic, first three rows (reference :75-83)'''
    extrinsic = np.array(cam[0], dtype=np.float64)
    intrinsic = np.array(cam[1], dtype=np.float64)
    intrinsic[3, :] = 0
    return np.matmul(intrinsic, extrinsic)[0:3]


def atvsnet_to_gipuma_cam(in_path, out_path):
    '''convert atvsnet camera to gipuma camera format (reference :70-93): 3 rows of `str(value) ` + blank line'''
    with open(in_path) as f:
        cam = load_cam(f)
    P = projection_matrix(cam)
    with open(out_path, "w") as f:
        for i in range(0, 3):
            for j in range(0, 4):
                f.write(str(P[i][j]) + ' ')
            f.write('\n')
        f.write('\n')


def fake_colmap_normal(in_depth_path, out_normal_path):
    '''(1,1,1)/sqrt(3) normals where the depth is positive, 0 elsewhere (reference :95-113)'''
    depth_image = read_gipuma_dmb(in_depth_path)
    h, w = depth_image.shape[:2]
    normal_image = np.ones((h, w, 3), np.float32) / 1.732050808
    mask = (depth_image > 0).astype(np.float32).reshape(h, w, 1)
    write_gipuma_dmb(out_normal_path, np.float32(normal_image * mask))


def atvsnet_to_gipuma(dense_folder, gipuma_point_folder):
    '''(reference :115-182)'''
    depth_folder = os.path.join(dense_folder, 'depths_atvsnet')
    gipuma_cam_folder = os.path.join(gipuma_point_folder, 'cams')
    gipuma_image_folder = os.path.join(gipuma_point_folder, 'images')
    for d in (gipuma_point_folder, gipuma_cam_folder, gipuma_image_folder):
        if not os.path.isdir(d):
            os.mkdir(d)
    image_names = sorted(os.listdir(depth_folder))
    for image_name in image_names:
        if not ('jpg' in image_name or 'png' in image_name):
            continue
        image_prefix = os.path.splitext(image_name)[0]
        in_cam_file = os.path.join(depth_folder, image_prefix + '.txt')
        atvsnet_to_gipuma_cam(in_cam_file, os.path.join(gipuma_cam_folder, image_name + '.P'))
        with open(in_cam_file) as f:
            cam = load_cam(f)
        intrinsic = np.array(cam[1])
        intrinsic[3, :] = 0
        np.save(os.path.join(gipuma_cam_folder, image_name[:-4] + '_intr.npy'), intrinsic)
        np.save(os.path.join(gipuma_cam_folder, image_name[:-4] + '_extr.npy'), np.array(cam[0]))
        np.save(os.path.join(gipuma_cam_folder, image_name[:-4] + '_proj.npy'), projection_matrix(cam))
    for image_name in image_names:
        if 'jpg' not in image_name:
            continue
        shutil.copy(os.path.join(depth_folder, image_name), os.path.join(gipuma_image_folder, image_name))
    gipuma_prefix = '2333__'
    for image_name in image_names:
        if not ('jpg' in image_name or 'png' in image_name):
            continue
        image_prefix = os.path.splitext(image_name)[0]
        sub_depth_folder = os.path.join(gipuma_point_folder, gipuma_prefix + image_prefix)
        if not os.path.isdir(sub_depth_folder):
            os.mkdir(sub_depth_folder)
        in_depth_pfm = os.path.join(depth_folder, image_prefix + '_prob_filtered.pfm')
        out_depth_dmb = os.path.join(sub_depth_folder, 'disp.dmb')
        atvsnet_to_gipuma_dmb(in_depth_pfm, out_depth_dmb)
        fake_colmap_normal(out_depth_dmb, os.path.join(sub_depth_folder, 'normals.dmb'))


def probability_filter(dense_folder, prob_threshold):
    '''depth := 0 where the probability map is under the threshold (reference :185-206)'''
    depth_folder = os.path.join(dense_folder, 'depths_atvsnet')
    for image_name in sorted(os.listdir(depth_folder)):
        if not ('jpg' in image_name or 'png' in image_name):
            continue
        image_prefix = os.path.splitext(image_name)[0]
        with open(os.path.join(depth_folder, image_prefix + '.pfm'), 'rb') as f:
            depth_map = load_pfm(f)
        with open(os.path.join(depth_folder, image_prefix + '_prob.pfm'), 'rb') as f:
            prob_map = load_pfm(f)
        depth_map = np.array(depth_map)
        depth_map[prob_map < prob_threshold] = 0
        write_pfm(os.path.join(depth_folder, image_prefix + '_prob_filtered.pfm'), depth_map)


# --------------------------------------------------------------------------- the fusibile executable, in process

def _rq(M):
    """M = K R, K upper triangular with a positive diagonal (cv::RQDecomp3x3 as decomposeProjectionMatrix uses it)."""
    Q, U = np.linalg.qr(np.flipud(M).T)
    K = np.flipud(np.fliplr(U.T))
    R = np.flipud(Q.T)
    S = np.diag(np.sign(np.diag(K)))
    return K @ S, S @ R


def pack_cameras(Ps):
    """(N,28) float32 per camera: P[12] | M_inv[9] | C[3] | P_col34[3] | f -- the fields of Camera_cu the kernel reads
    (fusibile/cameraGeometryUtils.h:377-433 with transformP = false, cam_scale = 1): P as read from the .P file,
    M_inv = inverse of its left 3x3, C = camera centre from the signed minors of P (:20-50), f = K[0,0] of its RQ
    decomposition."""
    out = np.zeros((len(Ps), 28), np.float32)
    for i, P in enumerate(Ps):
        P32 = np.asarray(P, np.float32).reshape(3, 4)
        P64 = P32.astype(np.float64)
        K, _ = _rq(P64[:, :3])
        det = lambda cols: np.linalg.det(P64[:, cols])      # noqa: E731
        C4 = np.array([det([1, 2, 3]), -det([0, 2, 3]), det([0, 1, 3]), -det([0, 1, 2])])
        out[i, 0:12] = P32.reshape(12)
        out[i, 12:21] = np.linalg.inv(P64[:, :3]).astype(np.float32).reshape(9)
        out[i, 21:24] = (C4[:3] / C4[3]).astype(np.float32)
        out[i, 24:27] = P32[:, 3]
        out[i, 27] = np.float32(K[0, 0] / K[2, 2])
    return out


def read_p_file(path):
    """3x4 projection matrix of a gipuma .P file (whitespace-separated, fusibile/fileIoUtils.h readPFileStrechaPmvs)."""
    with open(path) as f:
        vals = [float(v) for v in f.read().split()]
    return np.array(vals[:12], np.float64).reshape(3, 4)


def fuse_views(Ps, depths, normals, images_bgr, disp_thresh, normal_thresh, num_consistent, device=None):
    """runcuda + copy_point_cloud_to_host (fusibile/fusibile.cu:279-325, 422-427) on the MI355X: every camera in turn
    is the reference of one atvs_fusibile launch; its created points whose three coordinates are all non-zero are
    appended (camera-major, row-major).  -> (points (M,3) float32, colors (M,3) uint8 r,g,b)."""
    import torch
    from .. import ops
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
    cams = torch.from_numpy(pack_cameras(Ps)).to(dev)
    nd = np.concatenate([np.asarray(normals, np.float32), np.asarray(depths, np.float32)[..., None]], -1)
    img = np.asarray(images_bgr)
    img4 = np.concatenate([img.astype(np.float32), np.zeros(img.shape[:3] + (1,), np.float32)], -1)
    nd_d, img_d = torch.from_numpy(np.ascontiguousarray(nd)).to(dev), torch.from_numpy(np.ascontiguousarray(img4)).to(dev)
    pts, cols = [], []
    for ref in range(len(Ps)):
        coord, _, tex, created = ops.fusibile(cams, nd_d, img_d, ref, float(disp_thresh), float(normal_thresh),
                                              int(num_consistent))
        X = coord[..., :3]
        keep = (created > 0) & (X[..., 0] != 0) & (X[..., 1] != 0) & (X[..., 2] != 0)
        pts.append(X[keep].cpu().numpy())
        t = tex[keep].cpu().numpy()
        # (char)(int) of channels 2, 1, 0 of the averaged texture (fusibile/displayUtils.h:109-111)
        cols.append(np.stack([t[:, 2], t[:, 1], t[:, 0]], -1).astype(np.int32).astype(np.uint8))
    return np.concatenate(pts, 0), np.concatenate(cols, 0)


def _imread_bgr(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1].copy()


def depth_map_fusion(point_folder, fusibile_exe_path, disp_thresh, num_consistent):
    '''(reference :209-231 + fusibile/main.cpp:559-845) fuses the gipuma-format folder on the MI355X and writes
    <point_folder>/consistencyCheck-<date>-<time>/final3d_model.ply'''
    cam_folder = os.path.join(point_folder, 'cams')
    image_folder = os.path.join(point_folder, 'images')
    normal_thresh = 360            # degrees, as the reference passes it (:216)
    # sub-folders `<prefix>__<id>` whose name starts with '2' and has two underscores (main.cpp:619-649), sorted
    ids = []
    for sub in sorted(os.listdir(point_folder)):
        if not os.path.isdir(os.path.join(point_folder, sub)) or sub.count('_') < 2 or sub[0] != '2':
            continue
        first = sub.find('_') + 1
        ident = sub[first + sub[first:].find('_') + 1:]
        for ext in ('.png', '.jpg', '.ppm'):
            if os.path.exists(os.path.join(image_folder, ident + ext)):
                ids.append((sub, ident, ident + ext))
                break
    if not ids:
        raise RuntimeError('%s: no gipuma depth folders with matching images' % point_folder)
    Ps = [read_p_file(os.path.join(cam_folder, name + '.P')) for _, _, name in ids]
    images = np.stack([_imread_bgr(os.path.join(image_folder, name)) for _, _, name in ids], 0)
    depths = np.stack([read_gipuma_dmb(os.path.join(point_folder, sub, 'disp.dmb')) for sub, _, _ in ids], 0)
    normals = np.stack([read_gipuma_dmb(os.path.join(point_folder, sub, 'normals.dmb')) for sub, _, _ in ids], 0)
    if depths.shape[1:3] != images.shape[1:3]:
        raise RuntimeError('depth maps %s and images %s differ in size' % (depths.shape[1:3], images.shape[1:3]))
    pts, cols = fuse_views(Ps, depths, normals, images, disp_thresh, normal_thresh * np.pi / 180.0, int(num_consistent))
    out = os.path.join(point_folder, 'consistencyCheck-' + time.strftime('%Y%m%d-%H%M%S'))
    os.makedirs(out, exist_ok=True)
    print('Found %.2f million points' % (len(pts) / 1e6))
    ply.write_ply(os.path.join(out, 'final3d_model.ply'), pts, cols)
    return out


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--dense_folder', type=str, default='../eval/pointcloud/lakeside')
    parser.add_argument('--fusibile_exe_path', type=str, default='../fusibile/build/fusibile')
    parser.add_argument('--prob_threshold', type=float, default=0.8)
    parser.add_argument('--disp_threshold', type=float, default=0.01)
    parser.add_argument('--num_consistent', type=float, default=2)
    args = parser.parse_args(argv)
    dense_folder = args.dense_folder
    point_folder = os.path.join(dense_folder, 'points_atvsnet')
    if not os.path.isdir(point_folder):
        os.mkdir(point_folder)
    print('filter depth map with probability map')
    probability_filter(dense_folder, args.prob_threshold)
    print('Convert atvsnet output to gipuma input')
    atvsnet_to_gipuma(dense_folder, point_folder)
    print('Run depth map fusion & filter')
    depth_map_fusion(point_folder, args.fusibile_exe_path, args.disp_threshold, args.num_consistent)
    cloudlist = sorted(name for name in os.listdir(point_folder) if 'consistency' in name)
    shutil.copyfile(os.path.join(point_folder, cloudlist[-1], 'final3d_model.ply'),
                    os.path.join(dense_folder, 'final3d_model.ply'))


if __name__ == '__main__':
    main()
