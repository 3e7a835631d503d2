"""Model assembly of the reference's ``atvsnet/model.py`` on the HIP kernels.

Same function names, positional arguments, tensor layouts and return values as
/root/reference/atvsnet/model.py (cited per function).  Tensors are batch-first,
channel-last float32 device tensors with B = 1; ``cams`` is (B,N,2,4,4);
``depth_start`` / ``depth_interval`` are 1-element device tensors (example.py:66-69) and
are never read back to the host.

Differences from the reference, none of which changes a value:
* the D-unrolled Python loops of warps become one launch per volume;
* D-constant tensors (tf.tile of reference features, photo/geo errors) are written
  straight into their channel slice of the network input;
* optional keyword arguments let the caller pass feature-tower outputs it already has
  (the reference recomputes the reference tower for every source view).
"""
import os
import torch

from .. import ops
from ..cnn_wrapper.atvsnet import (AttAggregation, AttAggregation_keepchannel, AttAggregation_refine,
                                    AttAggregation_refine_keepchannel, CostVolRefineNet, OutputConv,
                                    OutputConv_refine, ResNetDS2SPP, ResNetDS2SPP_shallow_f16, StackedUNet,
                                    StackedUNet_prob)
from ..flags import AUTO_REUSE, FLAGS
from .homography_warping import (get_homographies, get_visual_hull, homography_warping, transform_depth_batch,          # noqa: F401
                                 homography_warping_by_depth, transform_depth)


def _scalar(t):
    return t.reshape(-1)[:1].contiguous()


def get_propability_map(cv, depth_map, depth_start, depth_interval):
    """Confidence of a depth estimate (reference :13-65): the sum of the probabilities of the 4 hypotheses around
    it.  cv (B,D,H,W) probability volume (already soft-maxed, as the reference passes it), depth_map (B,H,W,1)
    -> (B,H,W,1).  prob2depth(out_prob_map=True) uses the fused form (soft-max evaluated on the fly)."""
    if cv.shape[0] != 1:
        raise ValueError('get_propability_map: batch size must be 1 (FLAGS.batch_size)')
    D, H, W = cv.shape[1:4]
    p = ops.probability_map(cv[0].contiguous(), depth_map.reshape(H, W).contiguous(), _scalar(depth_start),
                            _scalar(depth_interval), up_scale=1, softmax=False)
    return p.reshape(1, H, W, 1)


def upsample_prob_vol(prob_vol, up_scale=4):
    """Bilinear x4 (align_corners) of every depth plane (reference :68-76): (B,D,h,w) -> (B,D,4h,4w).
    prob2depth_upsample does NOT call this: it regresses depth without materialising the volume."""
    B, D, h, w = prob_vol.shape
    # (D,h,w) viewed as (h,w,D) rows would need a transpose; resize plane by plane through channel slices
    out = torch.empty((D, h * up_scale, w * up_scale), dtype=torch.float32, device=prob_vol.device)
    for d in range(D):
        ops.resize_bilinear(prob_vol[0, d].reshape(h, w, 1), (h * up_scale, w * up_scale),
                            out=out[d].reshape(h * up_scale, w * up_scale, 1))
    return out.unsqueeze(0)


def prob2depth(prob_volume, depth_num, depth_start, depth_interval, out_prob_map=False):
    """Soft-argmin over the (inverse) depth hypotheses (reference :80-109): (B,D,H,W) -> (B,H,W,1)
    [, probability map (B,H,W,1) with out_prob_map=True: get_propability_map of softmax(-volume), :104-107]."""
    if prob_volume.shape[1] != depth_num:
        raise ValueError('prob2depth: volume has %d planes, depth_num is %d' % (prob_volume.shape[1], depth_num))
    vol = prob_volume[0].contiguous()
    ds, di = _scalar(depth_start), _scalar(depth_interval)
    d = ops.softargmin(vol, ds, di)
    depth = d.reshape(1, d.shape[0], d.shape[1], 1)
    if not out_prob_map:
        return depth
    p = ops.probability_map(vol, d, ds, di, up_scale=1, softmax=True)
    return depth, p.reshape(1, p.shape[0], p.shape[1], 1)


def prob2depth_upsample(prob_volume, depth_num, depth_start, depth_interval, out_prob_map=False):
    """(B,D,h,w) -> (depth (B,h,w,1), depth_up (B,4h,4w,1)) [, prob_map, prob_map_up] (reference :113-129): the x4
    bilinear upsampling of the pre-softmax cost, the soft-argmin and the probability gather never materialise
    the (D,4h,4w) volume."""
    ds, di = _scalar(depth_start), _scalar(depth_interval)
    vol = prob_volume[0].contiguous()
    up = ops.upsample_softargmin(vol, ds, di, 4)
    lo = ops.softargmin(vol, ds, di)
    depth, depth_up = lo.reshape(1, lo.shape[0], lo.shape[1], 1), up.reshape(1, up.shape[0], up.shape[1], 1)
    if not out_prob_map:
        return depth, depth_up
    p = ops.probability_map(vol, lo, ds, di, up_scale=1, softmax=True)
    p_up = ops.probability_map(vol, up, ds, di, up_scale=4, softmax=True)
    return depth, depth_up, p.reshape(1, p.shape[0], p.shape[1], 1), p_up.reshape(1, p_up.shape[0], p_up.shape[1], 1)


def output_conv(cost_volume, reuse=AUTO_REUSE):
    """(B,D,H,W,C) -> (B,D,H,W) (reference :132-135)."""
    tower = OutputConv({'data': cost_volume}, is_training=True, reuse=reuse)
    return tower.get_output().squeeze(-1)


def output_conv_refine(cost_volume, reuse=AUTO_REUSE):
    """(reference :137-140)."""
    tower = OutputConv_refine({'data': cost_volume}, is_training=True, reuse=reuse)
    return tower.get_output().squeeze(-1)


def extract_feature_shallow(images, ref_id=0, view_id=1, ref_feature=None):
    """Low-level features of the reference and the source image (reference :144-154)."""
    if ref_feature is None:
        ref_feature = ResNetDS2SPP_shallow_f16({'data': images[:, ref_id]}, is_training=True, reuse=AUTO_REUSE).get_output()
    view_feature = ResNetDS2SPP_shallow_f16({'data': images[:, view_id]}, is_training=True, reuse=AUTO_REUSE).get_output()
    return ref_feature, view_feature


def build_cost_volume(ref_feature, view_feature, cams, depth_num, depth_start, depth_interval, ref_id, view_id,
                      output_homo=False, warp_ref=False, lazy=False):
    """concat([tile(ref), stack_d warp_d(view)], channel) -> (B,D,H,W,2F) (reference :157-200).

    lazy=True returns an ops.SplitVolume (warped half (D,h,w,F) + the un-tiled reference features) that
    the networks consume without ever building the tiled half."""
    if warp_ref:
        raise NotImplementedError('build_cost_volume(warp_ref=True) is an unused branch of the reference (:175-184)')
    H = get_homographies(cams[:, ref_id], cams[:, view_id], depth_num=depth_num, depth_start=depth_start,
                         depth_interval=depth_interval)
    rf, vf = ref_feature[0].contiguous(), view_feature[0].contiguous()
    if lazy:
        F = rf.shape[-1]
        cv = ops.SplitVolume(ops.warp_planes(vf, H[0].contiguous()), rf,
                             [('c', i) for i in range(F)] + [('v', i) for i in range(F)])
    else:
        cv = ops.build_cost_volume(rf, vf, H[0]).unsqueeze(0)
    return (cv, H) if output_homo else cv


def cost_volume_reasoning(cost_volume, output_prob=True, output_filtered_cost=False, reuse=AUTO_REUSE):
    """Cost-volume regularisation (reference :204-223)."""
    if output_prob:
        tower = StackedUNet_prob({'data': cost_volume}, is_training=True, reuse=reuse)
        prob = tower.get_output().squeeze(-1)
        if output_filtered_cost:
            return prob, tower.get_output_by_name('conv_b2_6_1')
        return prob
    tower = StackedUNet({'data': cost_volume}, is_training=True, reuse=reuse)
    return tower.get_output_by_name('conv_b2_6_1')


def _cached_homographies(hom, key, left_cam, right_cam, D, depth_start, depth_interval):
    """get_homographies(...)[0] of the camera pair `key` = (left view id, right view id); `hom`: a dict shared by the stages of one
    depth map (the sweep of pair (0, v) is needed by the cost volume, the refinement volumes and the visual hull) or None."""
    if hom is not None and key is not None and key in hom:
        return hom[key]
    Hm = get_homographies(left_cam, right_cam, depth_num=D, depth_start=depth_start, depth_interval=depth_interval)[0].contiguous()
    if hom is not None and key is not None:
        hom[key] = Hm
    return Hm


def _refinement_volumes(b, bufs, init_ref, init_view, ref_cam, view_cam, hull_cam, ref_f, view_f, D, ds, di, depth_start,
                        depth_interval, hom=None, keys=(None, None), transformed=None):
    """The volume construction of reference :247-330 for ONE (reference, source) pair, written into sample b of the
    batched buffers `bufs` = (photo_var (B,D,h,w,F), photo_const (B,h,w,2F), geo_var (B,D,h,w,2), geo_const (B,h,w,2),
    vis_hull (B,D,h,w,1)).  init_ref / init_view (1,h,w,1); ref_f / view_f (1,h,w,F); cams (1,2,4,4).
    hull_cam: the camera get_visual_hull pairs the source's depth map with -- cams[:, id_reorder[1]], i.e. view 1
    whatever the current source is (quirk C6, homography_warping.py:343-352)."""
    photo_var, photo_const, geo_var, geo_const, hull = bufs
    h, w = init_ref.shape[1:3]
    chan = ref_f.shape[3]
    # transformed: (init_view in the reference camera, init_view in ... through the hull camera) computed by the caller for all
    # views at once (refinement_batch), else here
    init_view_trans = transformed[0] if transformed is not None else transform_depth(init_view, view_cam, ref_cam)
    Hm = _cached_homographies(hom, keys[0], ref_cam, view_cam, D, depth_start, depth_interval)
    rf, vf = ref_f[0].contiguous(), view_f[0].contiguous()
    dref = init_ref.reshape(h, w).contiguous()
    dvt = init_view_trans.reshape(h, w, 1).contiguous()
    rc, vc = ref_cam[0].contiguous(), view_cam[0].contiguous()
    # photo_group = [ |warp_d(view_f) - ref_f| * mask , tile(photo_err) , tile(ref_f) ]   (:270-280, 309-311, 329, 333)
    # geo_group   = [ geo_ref(1) , geo_view (mask tiled to `chan` identical channels, quirk C7) , tile(geo_err) ,
    #                 tile(init_ref) ]                                                    (:285-300, 313-316, 330, 334)
    # Only the D-varying channels are built as volumes; the tiled maps stay (h,w,C) (ops.SplitVolume).
    # photo_err / geo_err (:309-316): warp by the reference depth, |. - ref| * mask, straight into the tiled-channel buffers
    # ... together with the tiled maps that follow them (the reference features / the reference depth: copy_ref)
    ops.warp_by_depth_err(vf, rf, rc, vc, dref, photo_const[b], 0, 'bilinear', FLAGS.inverse_depth, copy_ref=True)
    ops.warp_by_depth_err(dvt, dref.reshape(h, w, 1), rc, vc, dref, geo_const[b], 0, 'nearest', FLAGS.inverse_depth, copy_ref=True)
    pieces = photo_var.dim() == 3            # (B, chan/8, planar_stride): chunk-planar fp16 pieces for the photo stem (ops.photo_pieces_ok)
    ops.warp_planes(vf, Hm, out=photo_var[b], mode=1, ref=rf, planar=pieces, pieces=pieces)    # (D,h,w,chan)
    ops.geo_volume(dref, dvt.reshape(h, w), Hm, ds, di, geo_var[b], 0, 1)      # geo_ref | geo_view: one launch, one 8-byte store per voxel
    # get_visual_hull(view_num = 2) (:321-324; homography_warping.py:329-387)
    if hull_cam is view_cam:
        h_hull, vt_hull = Hm, init_view_trans
    else:
        h_hull = _cached_homographies(hom, keys[1], ref_cam, hull_cam, D, depth_start, depth_interval)
        vt_hull = transformed[1] if transformed is not None else transform_depth(init_view, hull_cam, ref_cam)
    ops.visual_hull(dref, vt_hull.reshape(h, w).contiguous(), h_hull, ds, di, FLAGS.inverse_depth,
                    out=hull[b].reshape(D, h, w))
    return chan


def _hull_view(ref_id):
    """id_reorder[1] of get_visual_hull(view_num = 2) (homography_warping.py:343-346): the view whose CAMERA the hull
    uses for the second depth map."""
    ids = [0, 1]
    ids[0] = ref_id
    ids[ref_id] = 0
    return ids[1]


def _refine_net(bufs, prob_vol, chan, independent, residual_base=None):
    photo_var, photo_const, geo_var, geo_const, hull = bufs
    cmap = [('v', i) for i in range(chan)] + [('c', i) for i in range(2 * chan)]
    if photo_var.dim() == 3:
        photo = ops.SplitVolume(photo_var, photo_const, cmap, planar=tuple(geo_var.shape[1:4]), pieces=True)
    else:
        photo = ops.SplitVolume(photo_var, photo_const, cmap)
    geo = ops.SplitVolume(geo_var, geo_const, [('v', 0)] + [('v', 1)] * chan + [('c', 0), ('c', 1)])
    inputs = {'photo_group': photo, 'geo_group': geo, 'prob_vol': prob_vol, 'vis_hull': hull}
    if residual_base is not None:
        inputs['residual_base'] = residual_base
    tower = CostVolRefineNet(inputs, is_training=True, reuse=AUTO_REUSE, independent_samples=independent)
    out = (tower.get_output_by_name('global_refine_3dconv6_1'), tower.get_output().squeeze(-1))
    if residual_base is not None:
        out += (tower.get_output_by_name('global_refine_3dconv6_1_plus'),)
    return out


def _refinement_buffers(B, D, h, w, chan, like):
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=like.device)     # noqa: E731
    photo_var = e(B, chan // 8, ops.planar_stride(D, h, w)) if ops.photo_pieces_ok((D, h, w), chan) else e(B, D, h, w, chan)
    return (photo_var, e(B, h, w, 2 * chan), e(B, D, h, w, 2), e(B, h, w, 2), e(B, D, h, w, 1))


def refinement(init_depth_images, cams, depth_num, depth_start, depth_interval, images, prob_vol, ref_id, view_id,
               view_homographies=None, num_depths=None, depth_ref_id=None, depth_view_id=None,
               shallow_features=None):
    """Refinement network on photometric / geometric / visual-hull volumes (reference :227-339).

    init_depth_images (B,2,h,w,1), prob_vol (B,D,h,w) -> (cost residual (B,D,h,w,8), prob residual (B,D,h,w)).
    """
    if depth_ref_id is None:
        depth_ref_id = ref_id
    if depth_view_id is None:
        depth_view_id = view_id
    if num_depths is None:
        num_depths = FLAGS.view_num
    if num_depths != 2:
        raise NotImplementedError('refinement: num_depths=2 (every call site of the reference) is built')
    D = int(depth_num)
    ds, di = _scalar(depth_start), _scalar(depth_interval)
    init_ref = init_depth_images[:, depth_ref_id]             # (B,h,w,1)
    init_view = init_depth_images[:, depth_view_id]
    h, w = init_ref.shape[1:3]
    if shallow_features is None:
        ref_f, view_f = extract_feature_shallow(images, ref_id, view_id)
    else:
        ref_f, view_f = shallow_features
    bufs = _refinement_buffers(1, D, h, w, ref_f.shape[3], ref_f)
    view_cam = cams[:, view_id]
    hull_cam = view_cam if _hull_view(ref_id) == view_id else cams[:, _hull_view(ref_id)]
    chan = _refinement_volumes(0, bufs, init_ref, init_view, cams[:, ref_id], view_cam, hull_cam, ref_f, view_f, D, ds,
                               di, depth_start, depth_interval)
    return _refine_net(bufs, prob_vol.unsqueeze(-1), chan, False)


def refinement_batch(depth_ref, depth_views, prob_vol, cams, depth_num, depth_start, depth_interval, sources,
                     shallow, ref_id=0, shallow_index=None, hom=None, residual_base=None):
    """`refinement` of several source views against one reference estimate in ONE pass of the network
    (the reference calls it once per source, example.py:163-172): depth_ref (1,h,w,1), depth_views {source: (1,h,w,1)},
    prob_vol (1,D,h,w) shared, shallow (N,h,w,16) features of every view (shallow_index: {view id: row of shallow} when
    it holds a subset) -> (cost residuals (S,D,h,w,8), prob residuals (S,D,h,w)), S = len(sources), each sample with
    its own batch statistics.  cams are indexed by the view ids themselves.  hom: the depth map's homography cache
    (_cached_homographies), e.g. the one base_stage_batch filled.  residual_base (1,D,h,w,8): the cost volume the residuals are
    added to (TVSNet_refine, reference :439) -- a third result, residual_base + cost residual (S,D,h,w,8), then comes out of the
    pass that forms the residuals."""
    si = (lambda v: v) if shallow_index is None else (lambda v: shallow_index[v])
    D = int(depth_num)
    S = len(sources)
    ds, di = _scalar(depth_start), _scalar(depth_interval)
    h, w = depth_ref.shape[1:3]
    chan = shallow.shape[-1]
    bufs = _refinement_buffers(S, D, h, w, chan, shallow)
    # every source's depth map in the reference camera's frame, directly and through the hull camera (quirk C6): one launch
    jobs, where = [], []
    for v in sources:
        jobs.append((depth_views[v], cams[:, v], cams[:, ref_id]))
        where.append([len(jobs) - 1, None])
        if _hull_view(ref_id) != v:
            jobs.append((depth_views[v], cams[:, _hull_view(ref_id)], cams[:, ref_id]))
            where[-1][1] = len(jobs) - 1
    trans = transform_depth_batch(jobs)
    for b, v in enumerate(sources):
        view_cam = cams[:, v]
        hull_cam = view_cam if _hull_view(ref_id) == v else cams[:, _hull_view(ref_id)]
        tr = (trans[where[b][0]], trans[where[b][1]] if where[b][1] is not None else None)
        _refinement_volumes(b, bufs, depth_ref, depth_views[v], cams[:, ref_id], view_cam, hull_cam,
                            shallow[si(ref_id):si(ref_id) + 1], shallow[si(v):si(v) + 1], D, ds, di, depth_start,
                            depth_interval, hom, ((ref_id, v), (ref_id, _hull_view(ref_id))), tr)
    pv = ops.stack([prob_vol[0].unsqueeze(-1)] * S, 0) if S > 1 else prob_vol.unsqueeze(-1)      # the shared volume, once per sample
    return _refine_net(bufs, pv, chan, True, residual_base)


def feature_extraction_batch(images):
    """The 2-D feature tower of EVERY view in one pass: (1,N,H,W,3) -> (N,H/4,W/4,32); per-image statistics, equal to
    N calls of TVSNet_feature_extraction."""
    return ResNetDS2SPP({'data': images[0]}, is_training=True, reuse=AUTO_REUSE, independent_samples=True).get_output()


def shallow_feature_batch(images):
    """extract_feature_shallow's tower for every view in one pass: (1,N,H,W,3) -> (N,H/4,W/4,16)."""
    return ResNetDS2SPP_shallow_f16({'data': images[0]}, is_training=True, reuse=AUTO_REUSE,
                                    independent_samples=True).get_output()


def build_cost_volumes(features, cams, pairs, depth_num, depth_start, depth_interval, feature_index=None, hom=None):
    """build_cost_volume for several (reference view, source view) pairs as ONE SplitVolume of len(pairs) samples:
    features (N,h,w,F) of every view; pair (r, s) sweeps r's frustum and warps s's features into it (the depth range is
    the given one for every pair -- quirk C11: the reverse direction of a siamese pair uses the reference's too).
    feature_index: {view id: row of `features`} when `features` holds a subset of the views; cams are indexed by view id."""
    D = int(depth_num)
    N, h, w, F = features.shape
    B = len(pairs)
    fi = (lambda v: v) if feature_index is None else (lambda v: feature_index[v])
    # the warped half goes straight into the layout its one consumer reads best: chunk-planar (F/8, D, h, w, 8) for the
    # Winograd x-pair launch of conv_b0_0_1 | conv_b0_1_0 (dense 32-byte voxels per 8-channel chunk), else channel-last
    planar = ops.planar_cost_volume_ok((D, h, w), F)
    # ... and, for the split-operand x-pair kernel, as the two fp16 PIECES of every value: the warp splits once per value, the
    # consumer's staging wavefronts only move bytes (LDS-DMA)
    pieces = planar and ops.planar_pieces_ok((D, h, w), F)
    var = torch.empty((B, F // 8, ops.planar_stride(D, h, w)) if planar else (B, D, h, w, F), dtype=torch.float32,
                      device=features.device)
    for b, (r, s) in enumerate(pairs):
        Hm = _cached_homographies(hom, (r, s), cams[:, r], cams[:, s], D, depth_start, depth_interval)
        ops.warp_planes(features[fi(s)], Hm, out=var[b], planar=planar, pieces=pieces)
    const = ops.stack([features[fi(r)] for r, _ in pairs], 0) if B > 1 else features[fi(pairs[0][0]):fi(pairs[0][0]) + 1]
    return ops.SplitVolume(var, const.contiguous(), [('c', i) for i in range(F)] + [('v', i) for i in range(F)],
                           planar=(D, h, w) if planar else False, pieces=pieces)


def base_stage_batch(features, cams, depth_num, depth_start, depth_interval, fwd, rev, ref_i=0, feature_index=None, hom=None):
    """TVSNet_base_siamese for several source views in ONE pass of the regulariser (the reference runs it per source,
    example.py:144-149): `fwd` = sources whose reference->source direction is wanted (filtered cost volume, probability
    volume, depth), `rev` = sources whose source->reference direction is wanted (depth_view).
    -> (filtered (F,D,h,w,8), prob (F,D,h,w), depth_b2 (F,h,w,1), depth_view {source: (1,h,w,1)})."""
    D = int(depth_num)
    pairs = [(ref_i, v) for v in fwd] + [(v, ref_i) for v in rev]
    cv = build_cost_volumes(features, cams, pairs, D, depth_start, depth_interval, feature_index, hom)
    tower = StackedUNet_prob({'data': cv}, is_training=True, reuse=AUTO_REUSE, independent_samples=True)
    del cv
    prob = tower.get_output().squeeze(-1)                        # (B,D,h,w)
    filt = tower.get_output_by_name('conv_b2_6_1')               # (B,D,h,w,8)
    depth = ops.softargmin(prob, _scalar(depth_start), _scalar(depth_interval), groups=prob.shape[0])   # (B,h,w)
    nf = len(fwd)
    h, w = depth.shape[1:]
    depth_view = {v: depth[nf + i].reshape(1, h, w, 1) for i, v in enumerate(rev)}
    return filt[:nf], prob[:nf], depth[:nf].unsqueeze(-1), depth_view


def TVSNet_feature_extraction(images, view_i):
    """2-D feature tower of one image (reference :420-425): (B,N,H,W,3) -> (B,H/4,W/4,32)."""
    return ResNetDS2SPP({'data': images[:, view_i]}, is_training=True, reuse=AUTO_REUSE).get_output()


def TVSNet(images, cams, depth_num, depth_start, depth_interval, view_i, ref_i=0):
    """Two-view network (reference :346-377) -> refined_prob_vol (B,D,h,w)."""
    ref_feature = TVSNet_feature_extraction(images, ref_i)
    view_feature = TVSNet_feature_extraction(images, view_i)
    cost_vol_view = build_cost_volume(view_feature, ref_feature, cams, depth_num, depth_start, depth_interval,
                                      ref_id=view_i, view_id=0, lazy=True)
    prob_vol_view = cost_volume_reasoning(cost_vol_view, output_filtered_cost=False)
    del cost_vol_view
    depth_view = prob2depth(prob_vol_view, depth_num, depth_start, depth_interval)
    cost_vol = build_cost_volume(ref_feature, view_feature, cams, depth_num, depth_start, depth_interval, ref_id=0,
                                 view_id=view_i, lazy=True)
    prob_vol_b2, _ = cost_volume_reasoning(cost_vol, output_filtered_cost=True)
    del cost_vol
    depth_b2 = prob2depth(prob_vol_b2, depth_num, depth_start, depth_interval)
    init_depth_images = ops.stack([depth_b2, depth_view], dim=1)
    _, prob_residual = refinement(init_depth_images, cams, depth_num, depth_start, depth_interval, images, prob_vol_b2,
                                  ref_id=ref_i, view_id=view_i, view_homographies=None, num_depths=2, depth_ref_id=0,
                                  depth_view_id=1)
    return ops.add_n([prob_vol_b2[0], prob_residual[0]]).unsqueeze(0)


def TVSNet_base(images, cams, depth_num, depth_start, depth_interval, view_i, ref_i=0, ref_feature=None):
    """(reference :380-395) -> (depth_b2, prob_vol_b2, filtered_cost_volume)."""
    if ref_feature is None:
        ref_feature = TVSNet_feature_extraction(images, ref_i)
    view_feature = TVSNet_feature_extraction(images, view_i)
    cost_vol = build_cost_volume(ref_feature, view_feature, cams, depth_num, depth_start, depth_interval, ref_id=0,
                                 view_id=view_i, lazy=True)
    prob_vol_b2, filtered = cost_volume_reasoning(cost_vol, output_filtered_cost=True)
    depth_b2 = prob2depth(prob_vol_b2, depth_num, depth_start, depth_interval)
    return depth_b2, prob_vol_b2, filtered


def TVSNet_base_siamese(images, cams, depth_num, depth_start, depth_interval, view_i, ref_i=0, ref_feature=None,
                        side_stream=None, ref_ready=None):
    """Both directions of one (reference, source) pair (reference :398-417) ->
    (depth_b2, prob_vol_b2, filtered_cost_volume, depth_view).  Quirk C11: the reverse
    direction sweeps the reference camera's depth range.  The two directions only share the feature
    towers; with `side_stream` (a torch.cuda.Stream) the reverse one is issued there and joined before
    returning."""
    if ref_feature is None:
        ref_feature = TVSNet_feature_extraction(images, ref_i)
    view_feature = TVSNet_feature_extraction(images, view_i)
    if ref_ready is not None:        # ref_feature is being produced on another stream: needed from here on
        torch.cuda.current_stream(view_feature.device).wait_event(ref_ready)

    def reverse():
        cost_vol_view = build_cost_volume(view_feature, ref_feature, cams, depth_num, depth_start, depth_interval,
                                          ref_id=view_i, view_id=0, lazy=True)
        prob_vol_view = cost_volume_reasoning(cost_vol_view, output_filtered_cost=False, reuse=AUTO_REUSE)
        del cost_vol_view
        return prob2depth(prob_vol_view, depth_num, depth_start, depth_interval)

    cur = None
    if side_stream is not None and view_feature.is_cuda:
        cur = torch.cuda.current_stream(view_feature.device)
        side_stream.wait_stream(cur)
        with torch.cuda.stream(side_stream):
            depth_view = reverse()
    cost_vol = build_cost_volume(ref_feature, view_feature, cams, depth_num, depth_start, depth_interval, ref_id=0,
                                 view_id=view_i, lazy=True)
    prob_vol_b2, filtered = cost_volume_reasoning(cost_vol, output_filtered_cost=True)
    del cost_vol
    depth_b2 = prob2depth(prob_vol_b2, depth_num, depth_start, depth_interval)
    if cur is not None:
        cur.wait_stream(side_stream)
        if not torch.cuda.is_current_stream_capturing():     # a capturing graph owns its pool's lifetimes
            depth_view.record_stream(cur)
            view_feature.record_stream(side_stream)
            ref_feature.record_stream(side_stream)
    else:
        depth_view = reverse()
    return depth_b2, prob_vol_b2, filtered, depth_view


def TVSNet_refine(depth_b2, depth_view, prob_vol_b2, filtered_cost_volume, images, cams, depth_num, depth_start,
                  depth_interval, view_i, ref_i=0, shallow_features=None):
    """Refinement of one source view against the aggregated estimate (reference :428-441) ->
    (refined_prob_vol (B,D,h,w), refined_cost_volume (B,D,h,w,8))."""
    init_depth_images = ops.stack([depth_b2, depth_view], dim=1)
    cost_residual, prob_residual = refinement(init_depth_images, cams, depth_num, depth_start, depth_interval, images,
                                              prob_vol_b2, ref_id=ref_i, view_id=view_i, view_homographies=None,
                                              num_depths=2, depth_ref_id=0, depth_view_id=1,
                                              shallow_features=shallow_features)
    refined_cost_volume = ops.add_n([filtered_cost_volume[0], cost_residual[0]]).unsqueeze(0)
    refined_prob_vol = ops.add_n([prob_vol_b2[0], prob_residual[0]]).unsqueeze(0)
    return refined_prob_vol, refined_cost_volume


def cost_volume_aggregation(cost_volumes, reuse=AUTO_REUSE, keepchannel=False):
    """AAM1 (reference :445-456).  cost_volumes: (B,D,H,W,C,N-1), or a list of N-1 tensors (B,D,H,W,C)."""
    if keepchannel:
        return AttAggregation_keepchannel({'data': cost_volumes}, is_training=True, reuse=reuse).get_output()
    return AttAggregation({'data': cost_volumes}, is_training=True, reuse=reuse).get_output().squeeze(-1)


def cost_volume_aggregation_refine(cost_volumes, reuse=AUTO_REUSE, keepchannel=False):
    """AAM2 (reference :460-468)."""
    if keepchannel:
        return AttAggregation_refine_keepchannel({'data': cost_volumes}, is_training=True, reuse=reuse).get_output()
    return AttAggregation_refine({'data': cost_volumes}, is_training=True, reuse=reuse).get_output().squeeze(-1)
