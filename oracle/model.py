"""CPU restatement of the reference's model assembly and example drivers.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/atvsnet/model.py and the run loops of
/root/reference/atvsnet/example.py:140-186 (multi-view) / :265-272 (two-view).
"""
import torch

from . import homography_warping as G
from . import nets
from . import tf_ops as T


def upsample_prob_vol(prob_vol, up_scale=4):
    """model.py:68-76: bilinear x4 (align_corners) of each depth plane of the PRE-softmax cost."""
    B, D, h, w = prob_vol.shape
    x = prob_vol.permute(0, 2, 3, 1)
    x = T.resize_bilinear_align_corners(x, (h * up_scale, w * up_scale))
    return x.permute(0, 3, 1, 2).contiguous()


def get_propability_map(cv, depth_map, depth_start, depth_interval):
    """model.py:13-65: cv (B,D,H,W) probability volume, depth_map (B,H,W,1) -> (B,H,W,1) = the sum of the
    probabilities at clip(floor(d)), clip(floor(d))-1, clip(ceil(d)), clip(ceil(d))+1 (each clipped to [0, D-1];
    an integral d counts its plane twice), d = (depth - depth_start) / depth_interval.  Batch 1 (the
    reference's _repeat_ / meshgrid bookkeeping is the identity for FLAGS.batch_size = 1)."""
    B, D, H, W = cv.shape
    assert B == 1
    d = ((depth_map.reshape(-1) - depth_start[0]) / depth_interval[0])
    l0 = torch.clamp(torch.floor(d).to(torch.int64), 0, D - 1)
    l1 = torch.clamp(l0 - 1, 0, D - 1)
    r0 = torch.clamp(torch.ceil(d).to(torch.int64), 0, D - 1)
    r1 = torch.clamp(r0 + 1, 0, D - 1)
    flat = cv[0].reshape(D, H * W)
    pix = torch.arange(H * W)
    g = lambda idx: flat[idx, pix]          # noqa: E731
    return (g(l0) + g(l1) + g(r0) + g(r1)).reshape(1, H, W, 1)


def prob2depth(prob_volume, depth_num, depth_start, depth_interval, out_prob_map=False):
    """model.py:80-109 soft-argmin over (inverse) depth: (B,D,H,W) -> (B,H,W,1) [, probability map :104-107]."""
    B = prob_volume.shape[0]
    depth_end = depth_start + (float(depth_num) - 1.0) * depth_interval
    p = T.softmax(-1.0 * prob_volume, axis=1)
    soft = torch.stack([T.linspace(depth_start[i], depth_end[i], depth_num) for i in range(B)], 0)
    est = (soft.reshape(B, depth_num, 1, 1) * p).sum(dim=1).unsqueeze(3)
    if out_prob_map:
        return est, get_propability_map(p, est, depth_start, depth_interval)
    return est


def prob2depth_upsample(prob_volume, depth_num, depth_start, depth_interval, out_prob_map=False):
    """model.py:113-129 -> (depth (B,h,w,1), depth_up (B,4h,4w,1)) [, prob_map, prob_map_up]."""
    up = upsample_prob_vol(prob_volume)
    if out_prob_map:
        d_up, p_up = prob2depth(up, depth_num, depth_start, depth_interval, True)
        d, p = prob2depth(prob_volume, depth_num, depth_start, depth_interval, True)
        return d, d_up, p, p_up
    d_up = prob2depth(up, depth_num, depth_start, depth_interval)
    d = prob2depth(prob_volume, depth_num, depth_start, depth_interval)
    return d, d_up


def build_cost_volume(ref_feature, view_feature, cams, depth_num, depth_start, depth_interval, ref_id, view_id):
    """model.py:157-200: concat([tile(ref), stack_d warp(view, H_d)], channel) -> (B,D,h,w,2F)."""
    ref_cam = cams[:, ref_id]
    view_cam = cams[:, view_id]
    H = G.get_homographies(ref_cam, view_cam, depth_num, depth_start, depth_interval)
    warped = [G.homography_warping(view_feature, H[:, d]) for d in range(depth_num)]
    ref_tiled = ref_feature.unsqueeze(1).expand(-1, depth_num, -1, -1, -1)
    return torch.cat([ref_tiled, torch.stack(warped, 1)], dim=-1)


def cost_volume_reasoning(cost_volume, W, layers=None):
    """model.py:204-223 with output_prob=True: -> (prob_vol (B,D,h,w), conv_b2_6_1 (B,D,h,w,8))."""
    out, filt = nets.stacked_unet_prob(cost_volume, W, layers)
    return out.squeeze(-1), filt


def extract_feature_shallow(images, W, ref_id=0, view_id=1):
    """model.py:144-154."""
    return (nets.resnet_ds2_spp_shallow_f16(images[:, ref_id], W),
            nets.resnet_ds2_spp_shallow_f16(images[:, view_id], W))


def refinement_inputs(init_depth_images, cams, depth_num, depth_start, depth_interval, images, prob_vol, W,
                      ref_id, view_id, num_depths=2, depth_ref_id=0, depth_view_id=1, shallow=None):
    """The volume construction of model.py:227-336 (everything before CostVolRefineNet).

    returns dict(photo_group (B,D,h,w,48), geo_group (B,D,h,w,19), prob_vol (B,D,h,w,1),
    vis_hull (B,D,h,w,1)).
    """
    B = prob_vol.shape[0]
    prob_vol = prob_vol.unsqueeze(-1)
    init_ref = init_depth_images[:, depth_ref_id]            # (B,h,w,1)
    init_view = init_depth_images[:, depth_view_id]
    ref_cam, view_cam = cams[:, ref_id], cams[:, view_id]
    init_view_trans = G.transform_depth(init_view, view_cam, ref_cam)
    H = G.get_homographies(ref_cam, view_cam, depth_num, depth_start, depth_interval)
    if shallow is None:
        ref_f, view_f = extract_feature_shallow(images, W, ref_id, view_id)
    else:
        ref_f, view_f = shallow
    chan = ref_f.shape[3]
    dn = torch.tensor(float(depth_num), dtype=torch.float32)
    photo, geo_ref, geo_view = [], [], []
    for d in range(depth_num):
        wf, m = G.homography_warping(view_f, H[:, d], output_mask=True)
        photo.append(torch.abs(wf - ref_f) * m.to(ref_f.dtype).expand(-1, -1, -1, chan))
        val = (depth_start + float(d) * depth_interval).reshape(B, 1, 1, 1)
        itv = depth_interval.reshape(B, 1, 1, 1)
        geo_ref.append(torch.abs(init_ref - val) / itv / dn)
        wd, m2 = G.homography_warping(init_view_trans, H[:, d], output_mask=True)
        geo_view.append((torch.abs(wd - val) / itv / dn) * m2.to(init_ref.dtype).expand(-1, -1, -1, chan))
    cost_vol_photo = torch.stack(photo, 1)
    cost_vol_geo = torch.cat([torch.stack(geo_ref, 1), torch.stack(geo_view, 1)], dim=-1)
    wfeat, mp = G.homography_warping_by_depth(view_f, ref_cam, view_cam, init_ref, output_mask=True)
    photo_err = torch.abs(wfeat - ref_f) * mp.to(ref_f.dtype).expand(-1, -1, -1, chan)
    wdep, mg = G.homography_warping_by_depth(init_view_trans, ref_cam, view_cam, init_ref,
                                             output_mask=True, method='nearest')
    geo_err = torch.abs(wdep - init_ref) * mg.to(init_ref.dtype)

    def tile(x):
        return x.unsqueeze(1).expand(-1, depth_num, -1, -1, -1)
    vis_hull = G.get_visual_hull(init_depth_images.squeeze(-1), cams, depth_num, depth_start, depth_interval,
                                 ref_id=ref_id, view_num=num_depths)
    return {
        'photo_group': torch.cat([cost_vol_photo, tile(photo_err), tile(ref_f)], dim=-1),
        'geo_group': torch.cat([cost_vol_geo, tile(geo_err), tile(init_ref)], dim=-1),
        'prob_vol': prob_vol,
        'vis_hull': vis_hull,
    }


def refinement(init_depth_images, cams, depth_num, depth_start, depth_interval, images, prob_vol, W,
               ref_id, view_id, num_depths=2, depth_ref_id=0, depth_view_id=1, shallow=None):
    """model.py:227-339 -> (cost_residual (B,D,h,w,8), prob_residual (B,D,h,w))."""
    inp = refinement_inputs(init_depth_images, cams, depth_num, depth_start, depth_interval, images, prob_vol, W,
                            ref_id, view_id, num_depths, depth_ref_id, depth_view_id, shallow)
    out, c61 = nets.cost_vol_refine_net(inp['photo_group'], inp['geo_group'], inp['prob_vol'], inp['vis_hull'], W)
    return c61, out.squeeze(-1)


def TVSNet(images, cams, depth_num, depth_start, depth_interval, W, view_i, ref_i=0, stages=None):
    """model.py:346-377 (two-view): -> refined_prob_vol (B,D,h,w)."""
    S = {} if stages is None else stages
    ref_f = nets.resnet_ds2_spp(images[:, ref_i], W)
    view_f = nets.resnet_ds2_spp(images[:, view_i], W)
    S['ref_feature'], S['view_feature'] = ref_f, view_f
    cv_view = build_cost_volume(view_f, ref_f, cams, depth_num, depth_start, depth_interval, view_i, 0)
    pv_view, _ = cost_volume_reasoning(cv_view, W)
    depth_view = prob2depth(pv_view, depth_num, depth_start, depth_interval)
    cv = build_cost_volume(ref_f, view_f, cams, depth_num, depth_start, depth_interval, 0, view_i)
    S['cost_volume'] = cv
    pv_b2, filt = cost_volume_reasoning(cv, W)
    depth_b2 = prob2depth(pv_b2, depth_num, depth_start, depth_interval)
    S['prob_vol_b2'], S['filtered_cost_volume'], S['depth_b2'], S['depth_view'] = pv_b2, filt, depth_b2, depth_view
    init = torch.stack([depth_b2, depth_view], 1)
    _, prob_res = refinement(init, cams, depth_num, depth_start, depth_interval, images, pv_b2, W,
                             ref_id=ref_i, view_id=view_i, num_depths=2, depth_ref_id=0, depth_view_id=1)
    S['prob_residual'] = prob_res
    return pv_b2 + prob_res


def TVSNet_base_siamese(images, cams, depth_num, depth_start, depth_interval, W, view_i, ref_i=0, ref_feature=None):
    """model.py:398-417 -> (depth_b2, prob_vol_b2, filtered_cost_volume, depth_view)."""
    ref_f = nets.resnet_ds2_spp(images[:, ref_i], W) if ref_feature is None else ref_feature
    view_f = nets.resnet_ds2_spp(images[:, view_i], W)
    cv = build_cost_volume(ref_f, view_f, cams, depth_num, depth_start, depth_interval, 0, view_i)
    pv_b2, filt = cost_volume_reasoning(cv, W)
    depth_b2 = prob2depth(pv_b2, depth_num, depth_start, depth_interval)
    # quirk C11: the reverse direction uses the reference's depth_start / interval
    cv_view = build_cost_volume(view_f, ref_f, cams, depth_num, depth_start, depth_interval, view_i, 0)
    pv_view, _ = cost_volume_reasoning(cv_view, W)
    depth_view = prob2depth(pv_view, depth_num, depth_start, depth_interval)
    return depth_b2, pv_b2, filt, depth_view


def TVSNet_refine(depth_b2, depth_view, prob_vol_b2, filtered_cost_volume, images, cams, depth_num,
                  depth_start, depth_interval, W, view_i, ref_i=0, shallow=None):
    """model.py:428-441 -> (refined_prob_vol, refined_cost_volume)."""
    init = torch.stack([depth_b2, depth_view], 1)
    cost_res, prob_res = refinement(init, cams, depth_num, depth_start, depth_interval, images, prob_vol_b2, W,
                                    ref_id=ref_i, view_id=view_i, num_depths=2, depth_ref_id=0, depth_view_id=1,
                                    shallow=shallow)
    return prob_vol_b2 + prob_res, filtered_cost_volume + cost_res


def cost_volume_aggregation(cost_volumes, W):
    """AAM1, keepchannel=True (model.py:445-456; atvsnet.py:196-203)."""
    return nets.attention_aggregation(cost_volumes, W, 'attention_aggregate')


def cost_volume_aggregation_refine(cost_volumes, W):
    """AAM2, keepchannel=True (model.py:460-468; atvsnet.py:229-234)."""
    return nets.attention_aggregation(cost_volumes, W, 'attention_aggregate_refine')


def depth_start_interval(cams):
    """example.py:66-69: depth_start = cams[0,0,1,3,0], depth_interval = cams[0,0,1,3,1], shape (B,)."""
    return cams[:1, 0, 1, 3, 0].clone(), cams[:1, 0, 1, 3, 1].clone()


def run_twoview(images, cams, W, max_d, stages=None):
    """example.py:219-272 (graph + run): images (1,2,H,W,3), cams (1,2,2,4,4) ->
    inverse-depth map at full resolution (1,H,W,1) (before the host-side inversion)."""
    ds, di = depth_start_interval(cams)
    refined = TVSNet(images, cams, max_d, ds, di, W, view_i=1, ref_i=0, stages=stages)
    if stages is not None:
        stages['refined_prob_vol'] = refined
    _, depth_refined = prob2depth_upsample(refined, max_d, ds, di)
    return depth_refined


def run_multiview(images, cams, W, max_d, stages=None):
    """example.py:51-186 run order: base (per view) -> AAM1 -> refine (per view) -> AAM2 -> upsample."""
    S = {} if stages is None else stages
    n = images.shape[1]
    ds, di = depth_start_interval(cams)
    ref_f = nets.resnet_ds2_spp(images[:, 0], W)     # the reference recomputes this per view; same value
    filt, probs, dviews = [], [], []
    for v in range(1, n):
        _, pv, fc, dv = TVSNet_base_siamese(images, cams, max_d, ds, di, W, view_i=v, ref_i=0, ref_feature=ref_f)
        filt.append(fc)
        probs.append(pv)
        dviews.append(dv)
    cost_agg = cost_volume_aggregation(torch.stack(filt, -1), W)
    prob_agg = nets.output_conv(cost_agg, W, 'attention_prob_vol')
    depth_init = prob2depth(prob_agg, max_d, ds, di)
    S['filtered_cost_volumes'], S['depth_views'] = filt, dviews
    S['cost_volume_agg'], S['prob_volume_agg'], S['depth_agg_init'] = cost_agg, prob_agg, depth_init
    rcost = []
    for v in range(1, n):
        _, rc = TVSNet_refine(depth_init, dviews[v - 1], prob_agg, cost_agg, images, cams, max_d, ds, di, W,
                              view_i=v, ref_i=0)
        rcost.append(rc)
    rcost_agg = cost_volume_aggregation_refine(torch.stack(rcost, -1), W)
    rprob_agg = nets.output_conv(rcost_agg, W, 'attention_prob_vol_refine')
    S['refined_cost_volumes'], S['refined_cost_volume_agg'], S['refined_prob_volume_agg'] = rcost, rcost_agg, rprob_agg
    _, depth_refined = prob2depth_upsample(rprob_agg, max_d, ds, di)
    return depth_refined


def invert_depth(out, twoview):
    """example.py:183-186 (multi-view: < 1e-10 -> inf) / :269-272 (two-view: <= 0 -> inf)."""
    out = out.clone()
    if twoview:
        out[out <= 0] = float('inf')
    else:
        out[out < 1e-10] = float('inf')
    return 1.0 / out
