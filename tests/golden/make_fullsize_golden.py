"""Generates tests/golden/fullsize_<cfg>.npz: outputs of the CPU oracle (oracle/) at BASELINE.json's FULL
configurations on the seeded synthetic inputs and weights (the same ones bench.py uses), so that the
-m gpu tests can check the HIP path at the sizes the throughput is quoted on:

    cfg2    two-view 640x512, D=192                (BASELINE configs[1])   float32 oracle
    cfg2f64 the same with the oracle's NETWORKS evaluated in float64 (geometry stays float32 so the same
            pixels are sampled): the value both float32 pipelines approximate = the noise floor
    cfg3    5 views 640x512, D=192, multi-view     (BASELINE configs[2], the metric's configuration)
    cfg3s5  the same with 6 views = five sources (bench.py's `five_sources` line; both AANet modules run the 5-view form)
    cfg4    9 views 928x480, D=256, multi-view     (BASELINE configs[3]: 8 sources)
    cfg5    two-view 1600x1184, D=256              (BASELINE configs[4])
    cfg5f64 the float64-network evaluation of cfg5.  The plain oracle call keeps both 15.5 GB float64 cost volumes and every
            U-Net layer alive at once (more than the 62 GB of the build container: killed by the kernel); `twoview_f64(lean=True)`
            evaluates the SAME graph with the oracle's own layer functions but builds each cost volume in place and drops every
            volume after its last consumer (checked bit-for-bit against the plain call on a small scene before the big run)
    cfg5h   configs[4] at HALF the image size: two-view 800x576 (the network needs multiples of 32), D=256, the same camera model (synthetic.make_inputs)
    cfg5hf64 its float64-network evaluation: the noise floor of the configs[4]-shaped case (D=256, two-view, wide
            images) that fits this container -- the HIP path is asserted to be no further from it than 1.5 x the float32
            oracle is (tests/test_gpu_fullsize.py::test_cfg5_halfsize_against_the_float64_floor)

Each fixture holds the final full-resolution inverse-depth map plus a few intermediate maps / single depth
planes of stage volumes (a few MB).  Run from the repository root in the BUILD container (CPU only):

    python tests/golden/make_fullsize_golden.py cfg2 cfg2f64 cfg3 cfg4 cfg5

Measured here (8 cores): see the `seconds` entry each fixture records.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import atvsnet_amd                                     # noqa: E402,F401
from atvsnet_amd import synthetic, variables           # noqa: E402
from oracle import homography_warping as G             # noqa: E402
from oracle import model as OM                         # noqa: E402
from oracle import tf_ops as T                         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = {            # name: (views, H, W, D)
    'cfg2': (2, 512, 640, 192),
    'cfg3': (5, 512, 640, 192),
    'cfg3s5': (6, 512, 640, 192),        # the metric's configuration with FIVE source views ("also report 5-source N = 6")
    'cfg4': (9, 480, 928, 256),
    'cfg5': (2, 1184, 1600, 256),
    'cfg5h': (2, 576, 800, 256),
}


def weights():
    store = variables.VariableStore().init_synthetic(1234)
    return {k: torch.from_numpy(v) for k, v in store.host.items()}


def inputs(name):
    n, H, Wd, D = CONFIGS[name]
    imgs, cams = synthetic.make_inputs(n, H, Wd, D, seed=0)
    return torch.from_numpy(imgs), torch.from_numpy(cams), D


def twoview(name):
    imgs, cams, D = inputs(name)
    S = {}
    d = OM.run_twoview(imgs, cams, weights(), D, S)
    mid = D // 2
    return {'depth': d[0, ..., 0].numpy(), 'depth_b2': S['depth_b2'][0, ..., 0].numpy(),
            'depth_view': S['depth_view'][0, ..., 0].numpy(),
            'refined_prob_mid': S['refined_prob_vol'][0, mid].numpy(),
            'filtered_cost_mid': S['filtered_cost_volume'][0, mid].numpy()}


def lean_cost_volume(ref_feature, view_feature, cams, depth_num, depth_start, depth_interval, ref_id, view_id):
    """oracle.model.build_cost_volume written into ONE preallocated tensor (no list of planes + stack + cat)."""
    H = G.get_homographies(cams[:, ref_id], cams[:, view_id], depth_num, depth_start, depth_interval)
    B, h, w, F = ref_feature.shape
    cv = torch.empty((B, depth_num, h, w, 2 * F), dtype=ref_feature.dtype)
    cv[..., :F] = ref_feature.unsqueeze(1)
    for d in range(depth_num):
        cv[:, d, :, :, F:] = G.homography_warping(view_feature, H[:, d])
    return cv


def lean_stacked_unet_prob(box, W):
    """oracle.nets.stacked_unet_prob (cnn_wrapper/atvsnet.py:100-192) with the same layer functions in the same order;
    `box` = [cost volume] is emptied after the volume's two consumers and every layer is dropped after its last use."""
    from oracle import nets
    bf = 8
    L = {}
    for b in range(3):
        p = 'conv_b%d_' % b
        if b == 0:
            inp = box.pop()
        else:
            q = 'conv_b%d_' % (b - 1)
            inp = L.pop(q + '6_0') + L.pop(q + '0_1')
        L[p + '1_0'] = nets.conv_bn(inp, W, p + '1_0', bf * 2, 2)
        L[p + '0_1'] = nets.conv_bn(inp, W, p + '0_1', bf, 1)
        del inp
        L[p + '2_0'] = nets.conv_bn(L[p + '1_0'], W, p + '2_0', bf * 4, 2)
        L[p + '3_0'] = nets.conv_bn(L[p + '2_0'], W, p + '3_0', bf * 8, 2)
        if b == 0:
            i11, i21 = L.pop(p + '1_0'), L.pop(p + '2_0')
        else:
            i11 = L.pop(p + '1_0') + L.pop(q + '5_0')
            i21 = L.pop(p + '2_0') + L.pop(q + '4_0')
        L[p + '1_1'] = nets.conv_bn(i11, W, p + '1_1', bf * 2, 1)
        L[p + '2_1'] = nets.conv_bn(i21, W, p + '2_1', bf * 4, 1)
        del i11, i21
        L[p + '3_1'] = nets.conv_bn(L.pop(p + '3_0'), W, p + '3_1', bf * 8, 1)
        L[p + '4_0'] = nets.deconv_bn(L.pop(p + '3_1'), W, p + '4_0', bf * 4)
        if b == 0:
            i50 = L[p + '4_0'] + L[p + '2_1']
        else:
            i50 = L[p + '4_0'] + L.pop(p + '2_1') + L['conv_b0_2_1']
        L[p + '5_0'] = nets.deconv_bn(i50, W, p + '5_0', bf * 2)
        del i50
        if b == 0:
            i60 = L[p + '5_0'] + L[p + '1_1']
        else:
            i60 = L[p + '5_0'] + L.pop(p + '1_1') + L['conv_b0_1_1']
        L[p + '6_0'] = nets.deconv_bn(i60, W, p + '6_0', bf)
        del i60
    c61 = L.pop('conv_b2_6_0') + L.pop('conv_b2_0_1')
    L.clear()
    return nets.conv(c61, W, 'conv_b2_6_2', 1, 1, relu=False), c61


def lean_run_twoview(images, cams, W, max_d):
    """oracle.model.run_twoview / TVSNet (model.py:346-377, example.py:219-272) with nothing kept beyond its last use."""
    from oracle import nets
    ds, di = OM.depth_start_interval(cams)
    ref_f = nets.resnet_ds2_spp(images[:, 0], W)
    view_f = nets.resnet_ds2_spp(images[:, 1], W)
    box = [lean_cost_volume(view_f, ref_f, cams, max_d, ds, di, 1, 0)]
    pv_view, _ = lean_stacked_unet_prob(box, W)
    depth_view = OM.prob2depth(pv_view.squeeze(-1), max_d, ds, di)
    del pv_view, _
    box = [lean_cost_volume(ref_f, view_f, cams, max_d, ds, di, 0, 1)]
    del ref_f, view_f
    pv_b2, filt = lean_stacked_unet_prob(box, W)
    del filt
    pv_b2 = pv_b2.squeeze(-1)
    depth_b2 = OM.prob2depth(pv_b2, max_d, ds, di)
    init = torch.stack([depth_b2, depth_view], 1)
    _, prob_res = OM.refinement(init, cams, max_d, ds, di, images, pv_b2, W, ref_id=0, view_id=1, num_depths=2,
                                depth_ref_id=0, depth_view_id=1)
    refined = pv_b2 + prob_res
    del pv_b2, prob_res
    _, depth_refined = OM.prob2depth_upsample(refined, max_d, ds, di)
    return depth_refined


def twoview_f64(name, lean=False, scene=None):
    """The float64-network evaluation (see tests/golden/make_truth64_golden.py for the idea)."""
    imgs, cams, D = scene if scene is not None else inputs(name)
    W64 = {k: v.double() for k, v in weights().items()}

    def f32_boundary(fn):
        def wrapped(*args, **kw):
            args = [a.float() if isinstance(a, torch.Tensor) and a.dtype == torch.float64 else a for a in args]
            out = fn(*args, **kw)
            if isinstance(out, tuple):
                return tuple(o.double() if o.dtype == torch.float32 else o for o in out)
            return out.double() if out.dtype == torch.float32 else out
        return wrapped
    # torch's float64 conv3d on the CPU unfolds its input (27 x Cin doubles per voxel: 101 GB for the 64-channel cost volume
    # of cfg5h): large float64 convolutions run slab by slab along the depth axis -- the same sums, a bounded footprint
    plain_conv = T.conv

    def slab_conv(x, w, stride=1, padding='SAME', dilation=1, bias=None, explicit_pad=None):
        nsp = x.dim() - 2
        cost = x.numel() * 8.0 * float(np.prod(w.shape[:nsp]))
        if x.dtype != torch.float64 or nsp != 3 or cost < 6e9 or dilation != 1 or not isinstance(stride, int):
            return plain_conv(x, w, stride, padding, dilation, bias, explicit_pad)
        ks = w.shape[:3]
        if explicit_pad is not None:
            pads = list(explicit_pad)
        elif padding == 'SAME':
            pads = [T.same_pad(x.shape[1 + i], ks[i], stride)[:2] for i in range(3)]
        else:
            pads = [(0, 0)] * 3
        n = x.shape[1]
        out_d = (n + pads[0][0] + pads[0][1] - ks[0]) // stride + 1
        step = max(1, int(out_d * 6e9 / cost))
        outs = []
        for o0 in range(0, out_d, step):
            o1 = min(out_d, o0 + step)
            lo, hi = o0 * stride - pads[0][0], (o1 - 1) * stride - pads[0][0] + ks[0]
            sl = x[:, max(lo, 0):min(hi, n)]
            outs.append(plain_conv(sl, w, stride, 'VALID', 1, bias, [(max(0, -lo), max(0, hi - n)), pads[1], pads[2]]))
        return torch.cat(outs, 1)
    T.conv = slab_conv
    saved = {}
    for fn in ('homography_warping', 'homography_warping_by_depth', 'transform_depth', 'get_visual_hull'):
        saved[fn] = getattr(G, fn)
        setattr(G, fn, f32_boundary(saved[fn]))
    orig_linspace = T.linspace
    T.linspace = lambda a, b, n: orig_linspace(float(a), float(b), n).double()
    try:
        d64 = (lean_run_twoview if lean else OM.run_twoview)(imgs.double(), cams, W64, D)
    finally:
        for fn, f in saved.items():
            setattr(G, fn, f)
        T.linspace = orig_linspace
        T.conv = plain_conv
    return {'depth64': d64[0, ..., 0].numpy().astype(np.float32)}


def multiview(name):
    imgs, cams, D = inputs(name)
    S = {}
    d = OM.run_multiview(imgs, cams, weights(), D, S)
    mid = D // 2
    return {'depth': d[0, ..., 0].numpy(), 'depth_agg_init': S['depth_agg_init'][0, ..., 0].numpy(),
            'depth_views': torch.stack([v[0, ..., 0] for v in S['depth_views']], 0).numpy(),
            'cost_agg_mid': S['cost_volume_agg'][0, mid].numpy(),
            'refined_cost_agg_mid': S['refined_cost_volume_agg'][0, mid].numpy()}


def main(argv):
    torch.set_num_threads(int(os.environ.get('ORACLE_THREADS', min(os.cpu_count() or 1, 8))))
    for name in argv:
        t0 = time.time()
        with torch.no_grad():
            if name == 'cfg5f64':
                # the lean evaluation is the plain one, bit for bit (a small scene), before it is trusted at full size
                im, cm = synthetic.make_inputs(2, 128, 160, 32, seed=0)
                small = (torch.from_numpy(im), torch.from_numpy(cm), 32)
                a, b = twoview_f64(None, False, small)['depth64'], twoview_f64(None, True, small)['depth64']
                assert np.array_equal(a, b), 'lean float64 evaluation differs from the plain one'
                out = twoview_f64('cfg5', lean=True)
                out['lean_equals_plain_on_160x128x32'] = np.bool_(True)
            elif name.endswith('f64'):
                out = twoview_f64(name[:-3])
            elif CONFIGS[name][0] == 2:
                out = twoview(name)
            else:
                out = multiview(name)
        out['seconds'] = np.float64(time.time() - t0)
        out['threads'] = np.int64(torch.get_num_threads())
        path = os.path.join(HERE, 'fullsize_%s.npz' % name)
        np.savez_compressed(path, **out)
        print('%s: %.0f s, %.2f MB, depth mean %.6f' % (name, out['seconds'], os.path.getsize(path) / 1e6,
                                                       float(np.abs(out.get('depth', out.get('depth64'))).mean())),
              flush=True)


if __name__ == '__main__':
    main(sys.argv[1:] or ['cfg2', 'cfg2f64', 'cfg3'])
