"""ETH3D batch driver: depth + probability maps for every reference image of a set of scenes.

Mirror of /root/reference/atvsnet/eval_pointcloud.py (gen_data_list :60-94, load_data :97-203, run_eval_pc
:206-397, main :400-424): the same on-disk inputs (<scene>/pair.txt, images/%08d.jpg, cams/%08d_cam.txt) and
outputs (<savepath>/<scene>/depths_atvsnet/%08d.pfm, %08d_prob.pfm, %08d.jpg, %08d.txt, %08d.png,
zz_runtime.txt).  The session.run sequence of the reference (base per source, AAM1, refinement per source,
AAM2 with probability maps) is example.infer_multiview(out_prob_map=True), replayed from one HIP graph per
input shape.  Not available here and therefore not done: the optional ground-truth depth range from
depths/*.exr (:170-192, needs an EXR reader).
"""
from __future__ import print_function

import argparse
import os
import sys
import time

import numpy as np
import torch

from .. import variables
from ..flags import FLAGS
from ..tools.common import Notify
from . import example
from .preprocess import (center_image, crop_mvs_input, gen_pipeline_mvs_list, load_cam, scale_image, scale_mvs_camera,
                         scale_mvs_input, write_cam, write_pfm)

ETH3D_LOW_RES_TEST = ['lakeside', 'sand_box', 'storage_room', 'storage_room_2', 'tunnel']


def gen_data_list(dense_folder):
    """ mvs input path list (reference :60-94) """
    return gen_pipeline_mvs_list(dense_folder)


def _inverse_depth_range(cam):
    """Depth range row of a camera -> inverse-depth sweep (reference :155-170)."""
    depth_min, depth_interval = cam[1][3][0], cam[1][3][1]
    if cam[1][3][2] > 0 and cam[1][3][3] > 0:
        depth_max = cam[1][3][3]
    else:
        depth_max = depth_min + float(FLAGS.max_d - 1) * depth_interval
    disp_min, disp_max = 1.0 / depth_max, 1.0 / depth_min
    cam[1][3][:] = (disp_min, (disp_max - disp_min) / FLAGS.max_d, FLAGS.max_d, disp_max)


def _ground_truth_depth_range(ref_image, cams):
    """use ground truth depth range (reference :170-192): `<ref image>.txt` names the original image; when
    `<its folder with /images/ -> /depths/>/<name>.exr` exists, the (inverse-)depth sweep of EVERY view becomes
    [min, max] of that map over FLAGS.max_d hypotheses.  The EXR is read by tools/exr.py; an EXR that exists but cannot
    be decoded raises (a silently different sweep would change every depth map)."""
    note = ref_image[0:ref_image.rfind('.') + 1] + 'txt'
    if not os.path.exists(note):
        return
    with open(note, "r") as f:
        filename = f.readline()
    ref_image_path = ref_image[0:ref_image.rfind('/') + 1] + filename
    depth_path = ref_image_path.replace('/images/', '/depths/')
    depth_path = depth_path[0:depth_path.rfind('.') + 1] + 'exr'
    if not os.path.exists(depth_path):
        print(depth_path, 'not exist.')
        return
    from ..tools import exr
    depth_gt = exr.imread_first_channel(depth_path)
    if FLAGS.inverse_depth:
        depth_gt[depth_gt <= 0.0] = float("inf")
        depth_gt = 1.0 / (depth_gt)
    disp_max = np.max(depth_gt)
    depth_gt[depth_gt <= 0.0] = float("inf")
    disp_min = np.min(depth_gt)
    disp_interval = (disp_max - disp_min) / FLAGS.max_d
    for cam in cams:
        cam[1][3][0] = disp_min
        cam[1][3][1] = disp_interval
        cam[1][3][2] = FLAGS.max_d
        cam[1][3][3] = disp_max


def load_data(sample_list, data_index):
    """One pair.txt entry -> (scaled BGR images (1,N,h/4,w/4,3), centred images (1,N,h,w,3), cameras at
    sample_scale (1,N,2,4,4), a (1,h/4,w/4,1) placeholder, reference image index) (reference :97-203)."""
    data = sample_list[data_index]
    image_index = int(os.path.splitext(os.path.basename(data[0]))[0])
    found = len(data) // 2
    images, cams = [], []
    for view in range(FLAGS.view_num):
        src = view if view < found else 0            # missing sources are replaced by the reference view
        images.append(example._imread_bgr(data[2 * src]))
        with open(data[2 * src + 1]) as f:
            cam = load_cam(f, 1.0)
        if view < found and cam[1][3][2] == 0:
            cam[1][3][2] = FLAGS.max_d
        cams.append(cam)

    resize_scale = 1
    if FLAGS.adaptive_scaling:
        h_scale = max(float(FLAGS.max_h) / im.shape[0] for im in images)
        w_scale = max(float(FLAGS.max_w) / im.shape[1] for im in images)
        if h_scale > 1 or w_scale > 1:
            print("max_h, max_w should < W and H!")
            print(images[-1].shape, 'h_scale', h_scale, 'w_scale', w_scale)
            sys.exit(-1)
        resize_scale = max(h_scale, w_scale)
    images, cams = scale_mvs_input(images, cams, scale=resize_scale)
    images, cams = crop_mvs_input(images, cams, base_image_size=32)
    centered = [center_image(im) for im in images]
    if FLAGS.inverse_depth:
        for cam in cams:
            _inverse_depth_range(cam)
    _ground_truth_depth_range(data[0], cams)
    cams = scale_mvs_camera(cams, scale=FLAGS.sample_scale)
    scaled = [scale_image(im, scale=FLAGS.sample_scale) for im in images]
    scaled_depth = scaled[-1][:, :, 0:1].copy()
    return (np.stack(scaled, 0)[None], np.stack(centered, 0)[None], np.stack(cams, 0)[None], scaled_depth[None],
            image_index)


class _Pipelines(object):
    """One set of captured HIP graphs per input shape (scenes of one data set share it), SLOTS depth maps queued:
    submit() issues a depth map asynchronously, fetch() returns the oldest one's results as numpy arrays.
    CO_RESIDENT = False: the GPU runs one depth map at a time (the reference's scene loop, eval_pointcloud.py:291-396, is
    serial too); the second slot only lets the host load / submit the next view and write the previous one's files
    meanwhile.  True runs both maps' kernels concurrently (+4.5 % maps/s) and stays off until the co-residency fault of
    DESIGN.md appendix B is root-caused or a >= 10,000-map full-size soak is clean; 'cu_split' (cli --maps_in_flight cu_split)
    runs them concurrently on disjoint halves of every XCD, where that fault cannot occur."""

    SLOTS = 2
    CO_RESIDENT = False

    def __init__(self, device, use_graph=True):
        self.device, self.use_graph, self.cache = device, use_graph, {}
        self.pending = []            # (pipeline, ticket) or (None, tensors) in submission order

    def submit(self, images_data, cams_data):
        images = torch.from_numpy(np.ascontiguousarray(images_data, dtype=np.float32))
        cams = torch.from_numpy(np.ascontiguousarray(cams_data, dtype=np.float32))
        split = self.use_graph and self.CO_RESIDENT == 'cu_split'
        key = tuple(images.shape)
        if not (split and key in self.cache):
            # (cu_split: the slot's own stream copies the HOST tensors -- an upload on the default stream would wait for the map in
            # flight on the other half of the chip, example.PipelinedInference.submit; the first map of a shape still needs device
            # tensors to capture the graphs from)
            images, cams = images.to(self.device), cams.to(self.device)
        if not self.use_graph:
            # eager: computed here, with the drivers' range guard (an fp16-range overflow reruns the map on the fp32 kernels)
            self.pending.append((None, example.infer_checked(
                lambda: example.infer_multiview(images, cams, FLAGS.max_d, out_prob_map=True), self.device)))
            return
        p = self.cache.get(key)
        if p is None:
            p = self.cache[key] = example.PipelinedInference(images, cams, FLAGS.max_d, slots=self.SLOTS,
                                                                co_resident=self.CO_RESIDENT, out_prob_map=True)
        self.pending.append((p, p.submit(images, cams)))

    def room(self):
        return len(self.pending) < self.SLOTS

    def fetch(self):
        p, t = self.pending.pop(0)
        # result(): the fp32 rerun of a map whose split-operand replay overflowed; host=True: copied by the slot's own stream
        out = t if p is None else p.result(t, host=True)
        return [example.check_finite(o.cpu().numpy(), 'a network output') for o in out]

    def __call__(self, images_data, cams_data):
        self.submit(images_data, cams_data)
        return self.fetch()


def run_eval_pc(savepath, image_infos, use_graph=True):
    """(reference :206-397) image_infos: [[[dense_path, image_folder, scene_name], format], ...]"""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from PIL import Image
    assert FLAGS.view_num > 2, 'the ETH3D driver runs the multi-view (AANet) pipeline'
    example._load_weights()
    torch.cuda.set_device(FLAGS.gpu_id)          # every kernel launches on the current device's stream
    run = _Pipelines(torch.device('cuda:%d' % FLAGS.gpu_id), use_graph)
    for image_info, _fmt in image_infos:
        mvs_list = gen_data_list(image_info[0])
        savepath_current = os.path.join(savepath, image_info[2])
        output_folder = os.path.join(savepath_current, 'depths_atvsnet')
        os.makedirs(output_folder, exist_ok=True)
        scene_runtime = 0.0
        start_time = time.time()
        queued = []                  # host-side data of the depth maps in flight, in submission order

        def finish():
            image_data_raw, cams_data, out_index = queued.pop(0)
            depth, depth_up, prob, prob_up = run.fetch()
            disp_up = np.squeeze(depth_up.copy())
            if FLAGS.inverse_depth:
                for m in (depth, depth_up):
                    m[m <= 0] = float("inf")
                depth, depth_up = 1.0 / depth, 1.0 / depth_up
            stem = os.path.join(output_folder, '%08d' % out_index)
            write_pfm(stem + '.pfm', np.squeeze(depth).astype(np.float32))
            write_pfm(stem + '_prob.pfm', np.squeeze(prob).astype(np.float32))
            if getattr(FLAGS, 'write_upsampled', False):      # commented out in the reference (:378-379)
                write_pfm(stem + '_up.pfm', np.squeeze(depth_up).astype(np.float32))
                write_pfm(stem + '_prob_up.pfm', np.squeeze(prob_up).astype(np.float32))
            Image.fromarray(np.ascontiguousarray(image_data_raw[0, 0][:, :, ::-1])).save(stem + '.jpg')
            write_cam(stem + '.txt', cams_data[0, 0])
            plt.imsave(stem + '.png', disp_up, cmap='viridis')

        # the depth maps of a scene are independent: the next one is submitted before the previous one's results are
        # fetched and written, so file I/O overlaps the GPU (which still runs one map at a time: _Pipelines.CO_RESIDENT)
        for current_i in range(len(mvs_list)):
            image_data_raw, images_data, cams_data, _depth, out_index = load_data(mvs_list, current_i)
            if not run.room():
                finish()
            run.submit(images_data, cams_data)
            queued.append((image_data_raw, cams_data, out_index))
        while queued:
            finish()
        scene_runtime = time.time() - start_time       # wall clock of the scene (the reference sums sess.run times)
        with open(os.path.join(savepath_current, 'zz_runtime.txt'), "w") as text_file:
            text_file.write('runtime ' + str(scene_runtime))
        print(Notify.INFO, '%s: %d depth maps, %.2f s' % (image_info[2], len(mvs_list), scene_runtime), Notify.ENDC)


def main(scene_list=None, base_path='eth3d/'):
    """(reference :400-424)"""
    scene_list = ETH3D_LOW_RES_TEST if scene_list is None else scene_list
    os.makedirs(FLAGS.savepath, exist_ok=True)
    FLAGS.max_h = int(FLAGS.max_h / 32) * 32
    FLAGS.max_w = int(FLAGS.max_w / 32) * 32
    image_infos = []
    for scene in scene_list:
        folder = os.path.join(FLAGS.data_root, base_path + scene)
        image_infos.append([[folder, os.path.join(folder, 'images'), scene], 'preprocessed'])
    run_eval_pc(FLAGS.savepath, image_infos, use_graph=not getattr(FLAGS, 'eager', False))


def cli(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--data_root', type=str, default=FLAGS.data_root)
    parser.add_argument('--savepath', type=str, default=FLAGS.savepath)
    parser.add_argument('--pretrained_model_ckpt_path', type=str, default=FLAGS.pretrained_model_ckpt_path)
    parser.add_argument('--view_num', type=int, default=8)
    parser.add_argument('--max_d', type=int, default=128)
    parser.add_argument('--max_w', type=int, default=FLAGS.max_w)
    parser.add_argument('--max_h', type=int, default=FLAGS.max_h)
    parser.add_argument('--gpu_id', type=int, default=FLAGS.gpu_id)
    parser.add_argument('--scenes', type=str, default=None, help='comma-separated scene folders under data_root/eth3d/')
    parser.add_argument('--synthetic_weights', action='store_true')
    parser.add_argument('--write_upsampled', action='store_true')
    parser.add_argument('--eager', action='store_true')
    parser.add_argument('--maps_in_flight', choices=('serial', 'cu_split'), default='serial',
                        help='serial: one depth map on the GPU at a time (default); cu_split: the two queued depth maps run concurrently, '
                             'each on its own half of every XCD (example.cu_split_streams: no SIMD shared between them; same bits, '
                             'more depth maps per second, twice the latency of one)')
    args = parser.parse_args(argv)
    scenes = args.scenes.split(',') if args.scenes else None
    _Pipelines.CO_RESIDENT = 'cu_split' if args.maps_in_flight == 'cu_split' else False
    for k, v in vars(args).items():
        if k not in ('scenes', 'maps_in_flight'):
            setattr(FLAGS, k, v)
    print('Evaluate A-TVSNet pointcloud with %d views' % (FLAGS.view_num))
    main(scenes)


if __name__ == '__main__':
    cli()
