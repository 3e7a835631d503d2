"""Source views sharded across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The base stage
(towers, warp, 2x U-Net, soft-argmin) and the refinement stage are independent per source
view (reference example.py:144-149, 163-172): view v runs on rank (v-1) mod G with
replicated weights and reference features (with >= 2 ranks per source the two siamese directions of a
pair split over two ranks, see plan()); training-mode BN statistics are per view call, so no BN
collective exists.  The only exchange is inside the two AANet modules
(reference network.py:282-351, 378-408), split at their three reductions over views:

    S_sum  = all_reduce_SUM( sum_local S_n )
    U_max  = all_reduce_MAX( max_local (R_n - S_n + S_sum) )
    [den, num] = all_reduce_SUM( [sum_local e_n, sum_local e_n * X_n] ),  e_n = exp(U_n - U_max)
    out = num / den            (then the 8->1 output conv and soft-argmin, replicated)

so every rank ends each AAM with the aggregated volume its refinement needs.  Messages are
V*8 fp32 (126 MB at 160x128x192) -- large enough that RCCL's ring runs at link rate.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import ops
from . import variables
from .flags import FLAGS


class HipAttentionOps(object):
    """The local pieces of the sharded AANet on the HIP kernels."""

    def scores(self, xs, scope):
        st = variables.default_store()
        wu = st.get_host('%s/attention_activation/weight_unique' % scope, (3, 3, 3, 8, 8))
        ws = st.get_host('%s/attention_activation/weight_shared' % scope, (3, 3, 3, 8, 8))
        w16 = np.concatenate([ws, wu], axis=-1)
        return [ops.conv(x, scope + '/attention_activation/shared|unique', w16, relu=True) for x in xs]

    def partial(self, srs, xs, stage, ssum=None, umax=None):
        return ops.aanet_partial(srs, xs, stage, ssum=ssum, umax=umax)

    def divide(self, num, den):
        return ops.divide(num, den)


def _attention_steps(local_xs, scope, like, impl=None):
    """Generator form of the sharded AANet: yields (tensor, reduce op) wherever the ranks must all-reduce that
    tensor in place, and returns the aggregated (D,h,w,8) volume (identical on every rank)."""
    impl = impl or HipAttentionOps()
    shape = tuple(like.shape)
    if local_xs:
        srs = impl.scores(local_xs, scope)
        ssum = impl.partial(srs, local_xs, 0)
    else:
        ssum = torch.zeros(shape, dtype=torch.float32, device=like.device)
    yield ssum, dist.ReduceOp.SUM
    if local_xs:
        umax = impl.partial(srs, local_xs, 1, ssum=ssum)
    else:
        umax = torch.full(shape, float('-inf'), dtype=torch.float32, device=like.device)
    yield umax, dist.ReduceOp.MAX
    if local_xs:
        acc = impl.partial(srs, local_xs, 2, ssum=ssum, umax=umax)
    else:
        acc = torch.zeros((2,) + shape, dtype=torch.float32, device=like.device)
    yield acc, dist.ReduceOp.SUM
    return impl.divide(acc[1], acc[0])


def _drive(gen, group=None):
    """Run a step generator eagerly: perform every yielded all-reduce, return the generator's result."""
    try:
        while True:
            tensor, op = next(gen)
            dist.all_reduce(tensor, op=op, group=group)
    except StopIteration as e:
        return e.value


def sharded_attention(local_xs, scope, like, impl=None, group=None):
    """AANet over views that live on different ranks.

    local_xs: this rank's list of (D,h,w,8) tensors (may be empty); `like`: a tensor giving the
    (D,h,w,8) shape / device for ranks that own no view.  Returns the aggregated (D,h,w,8) volume,
    identical on every rank.
    """
    return _drive(_attention_steps(local_xs, scope, like, impl), group)


EXCHANGE = 'three all-reduces per AANet: SUM S, MAX U, SUM [e, e*X]'


def rank_groups(num_views, world, split_directions=False):
    """Ranks -> groups; every group computes its own depth map with that map's source views sharded over its ranks.

    A group never has more ranks than the depth map has source views (one source per GPU is the finest useful
    partition: both siamese directions, the refinement and the AANet share of a source stay on its owner, so no rank
    idles through a stage); ranks beyond that form further groups that work on further depth maps (the other
    reference views of a scene are independent, reference eval_pointcloud.py:399-424).  split_directions=True keeps
    ONE group and deals the two directions of a pair to two ranks instead (plan()).
    """
    nsrc = max(num_views - 1, 1)
    if split_directions or world <= nsrc:
        return [list(range(world))]
    return [list(range(a, min(a + nsrc, world))) for a in range(0, world, nsrc)]


def local_views(num_views, rank, world):
    """Source views (1..N-1) whose refinement (and forward direction) `rank` owns."""
    return [v for (kind, v) in plan(num_views, world)[rank] if kind == 'fwd']


def plan(num_views, world):
    """Work of every rank: a list of ('fwd', v) / ('rev', v) tasks per rank.

    'fwd' = reference->source direction of the base stage (filtered cost volume) AND the later refinement
    of source v; 'rev' = source->reference direction (depth_view only).  With at least two ranks per source
    the two directions of a pair run on different ranks (the (h,w) depth_view map travels in one tiny
    all-reduce); otherwise sources are dealt round-robin with both directions on the owner.
    """
    nsrc = num_views - 1
    tasks = [[] for _ in range(world)]
    if world >= 2 * nsrc:
        for v in range(1, num_views):
            tasks[2 * (v - 1)].append(('fwd', v))
            tasks[2 * (v - 1) + 1].append(('rev', v))
    else:
        g = min(world, nsrc)
        for v in range(1, num_views):
            tasks[(v - 1) % g] += [('fwd', v), ('rev', v)]
    return tasks


def _sharded_steps(images, cams, max_d, world, rank, view_streams=True):
    """example.infer_multiview for this rank's share of the source views, as a generator: local compute runs
    between the yields, every yield is (tensor, reduce op) = an in-place all-reduce all ranks must perform.
    Returns the full-resolution inverse-depth map (1,H,W,1), identical on every rank."""
    from .atvsnet import example as ex
    from .atvsnet import model
    from .cnn_wrapper.atvsnet import ResNetDS2SPP_shallow_f16
    n = images.shape[1]
    mine = plan(n, world)[rank]
    fwd = [v for kind, v in mine if kind == 'fwd']
    rev = [v for kind, v in mine if kind == 'rev']
    views = sorted(set(fwd + rev))
    depth_start, depth_interval = ex.depth_range(cams)
    dev = images.device
    h, w = images.shape[2] // 4, images.shape[3] // 4
    vs = ex._ViewStreams(len(views), dev, view_streams)          # one stream per owned source view
    slot = {v: i for i, v in enumerate(views)}
    ref_feature = model.TVSNet_feature_extraction(images, 0) if mine else None
    dv_all = torch.zeros((n - 1, h, w), dtype=torch.float32, device=dev)

    # ---- base stage (reference model.py:398-417): the owned directions of every owned view
    def base(v):
        feat = model.TVSNet_feature_extraction(images, v)
        f = None
        if v in fwd:
            cv = model.build_cost_volume(ref_feature, feat, cams, max_d, depth_start, depth_interval, ref_id=0,
                                         view_id=v, lazy=True)
            f = model.cost_volume_reasoning(cv, output_filtered_cost=True)[1][0]
        if v in rev:      # quirk C11: sweeps the reference camera's depth range
            cv = model.build_cost_volume(feat, ref_feature, cams, max_d, depth_start, depth_interval, ref_id=v,
                                         view_id=0, lazy=True)
            dv = model.prob2depth(model.cost_volume_reasoning(cv, output_filtered_cost=False), max_d, depth_start,
                                  depth_interval)
            dv_all[v - 1].copy_(dv.reshape(h, w))
        return f
    outs = [vs.run(slot[v], lambda v=v: base(v)) for v in views]
    vs.join(outs)
    filtered = [o for o in outs if o is not None]
    del outs
    yield dv_all, dist.ReduceOp.SUM                 # every view has exactly one contributor

    # ---- AAM1 across ranks
    like = torch.empty((max_d, h, w, 8), dtype=torch.float32, device=dev) if not filtered else filtered[0]
    cost_agg = (yield from _attention_steps(filtered, 'attention_aggregate', like)).unsqueeze(0)
    prob_agg = model.output_conv(cost_agg, reuse=False)
    depth_init = model.prob2depth(prob_agg, max_d, depth_start, depth_interval)
    del filtered

    # ---- refinement of the owned sources
    ref_shallow = ResNetDS2SPP_shallow_f16({'data': images[:, 0]}, is_training=True).get_output() if fwd else None

    def refine(v):
        shallow = model.extract_feature_shallow(images, 0, v, ref_feature=ref_shallow)
        return model.TVSNet_refine(depth_init, dv_all[v - 1].reshape(1, h, w, 1), prob_agg, cost_agg, images, cams,
                                   max_d, depth_start, depth_interval, view_i=v, ref_i=0, shallow_features=shallow)[1][0]
    refined = [vs.run(slot[v], lambda v=v: refine(v)) for v in fwd]
    vs.join(refined)

    # ---- AAM2 across ranks, head, x4 upsample + soft-argmin (replicated)
    rcost_agg = yield from _attention_steps(refined, 'attention_aggregate_refine', like)
    rprob_agg = model.output_conv_refine(rcost_agg.unsqueeze(0), reuse=False)
    _, depth_refined = model.prob2depth_upsample(rprob_agg, max_d, depth_start, depth_interval)
    return depth_refined


def infer_multiview_sharded(images, cams, max_d=None, group=None, view_streams=True):
    """example.infer_multiview with the source views sharded over the process group (every launch issued from
    Python).  Every rank returns the same full-resolution inverse-depth map (1,H,W,1)."""
    max_d = FLAGS.max_d if max_d is None else max_d
    return _drive(_sharded_steps(images, cams, max_d, dist.get_world_size(group), dist.get_rank(group), view_streams),
                  group)


class ShardedGraphedInference(object):
    """The sharded pipeline as a chain of HIP graphs with the collectives between them.

    The local compute between two all-reduces is captured once into a HIP graph (per-view streams forked and
    joined inside it); the all-reduces themselves stay ordinary RCCL calls on the tensors the graphs own.  One
    step = replay, all-reduce, replay, ... (7 all-reduces, 8 graphs) instead of ~500-1000 launches issued from
    Python per rank, which is what bounds the eager sharded path.  All graphs share one memory pool and are
    replayed in capture order.  Inputs live in static buffers: pass new images / cams to __call__."""

    def __init__(self, images, cams, max_d=None, group=None, view_streams=True):
        self.max_d = FLAGS.max_d if max_d is None else max_d
        self.group = group
        self.images, self.cams = images.clone(), cams.clone()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = images.device
        # warm-up on a side stream: weight packing / uploads, function attributes, communicator set-up
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            _drive(_sharded_steps(self.images, self.cams, self.max_d, world, rank, view_streams), group)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graphs, self.colls = [], []
        pool = torch.cuda.graph_pool_handle()
        gen = _sharded_steps(self.images, self.cams, self.max_d, world, rank, view_streams)
        done = False
        while not done:
            g = torch.cuda.CUDAGraph()
            item = None
            # thread_local: the RCCL watchdog thread polls its events (hipEventQuery) while this thread captures;
            # in the default global mode that call is "not permitted when stream is capturing" and aborts the process
            with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                try:
                    item = next(gen)
                except StopIteration as e:
                    self.out, done = e.value, True
            self.graphs.append(g)
            if not done:
                # during capture nothing ran: the tensor holds whatever the pool had; the all-reduce keeps the ranks'
                # collective sequences aligned and is repeated, on real data, in every step
                dist.all_reduce(item[0], op=item[1], group=group)
                self.colls.append(item)
        torch.cuda.synchronize(dev)
        self._weights = (ops.cache_snapshot(), variables.default_store().device_snapshot())

    def __call__(self, images=None, cams=None):
        if images is not None:
            self.images.copy_(images)
        if cams is not None:
            self.cams.copy_(cams)
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.colls):
                dist.all_reduce(self.colls[i][0], op=self.colls[i][1], group=self.group)
        return self.out
