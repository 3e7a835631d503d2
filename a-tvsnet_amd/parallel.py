"""Source views sharded across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The ranks form GROUPS of at most one
rank per source view (rank_groups); a group computes one depth map, further groups compute further depth maps
(the reference views of a scene are independent, reference eval_pointcloud.py:399-424).

Inside a group the base stage (towers, warp, 2x U-Net, soft-argmin) and the refinement stage are independent per
source view (reference example.py:144-149, 163-172): view v runs on rank (v-1) mod G with replicated weights and its
own copy of the reference tower; training-mode BN statistics are per view call, so no BN collective exists.  A rank
evaluates every network ONCE over all the (view, direction) calls it owns (model.*_batch).

The only exchange is inside the two AANet modules (reference network.py:282-351, 378-408), whose softmax runs
over the views at every voxel.  Voxel-transposed form:

    1. every rank computes SR_n = [relu(conv(X_n, W_shared)) | relu(conv(X_n, W_unique))] for its views;
    2. ALL-TO-ALL (point-to-point sends in one RCCL group call): rank r receives rows [r*S, (r+1)*S) of every view's
       SR_n (16 ch) and X_n (8 ch) -- the row ranges of a channel-last volume are contiguous, nothing is packed;
    3. the ordinary atvs_aanet_combine on its V/G voxels over ALL views in view order (bit-identical to the
       single-GPU kernel on those voxels);
    4. ALL-GATHER of the (S, 8) results: every rank ends with the aggregated (D,h,w,8) volume its refinement needs.

Per rank and AANet: send (G-1)/G * n_local * 24 * V * 4 B, receive the same for the other ranks' views plus
(G-1)/G * 8 * V * 4 B in the all-gather -- half the bytes of the three-all-reduce form (SUM S, MAX U, SUM [e, eX]),
every xGMI link busy at once, and no second pass over the local volumes.  The X_n rows are sent BEFORE the 8->16
convolution is issued, so their transfer overlaps it.
"""
import numpy as np
import warnings

import torch
import torch.distributed as dist

from . import ops
from . import variables
from .flags import FLAGS

EXCHANGE = 'all-to-all of voxel shards of [S|R|X] + all-gather of the combined shard, per AANet'


# --------------------------------------------------------------------------------------------- partition

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
    """Work of every rank of a group: a list of ('fwd', v) / ('rev', v) tasks per rank.

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


def owner_of(num_views, world):
    """{source view: rank that owns its forward direction (and so its AANet input)}."""
    return {v: r for r, tasks in enumerate(plan(num_views, world)) for kind, v in tasks if kind == 'fwd'}


# --------------------------------------------------------------------------------------------- transport

def _is_nccl(group):
    return dist.get_backend(group) == 'nccl'


def _p2p(sends, recvs, group):
    """Post every (tensor, peer) send and receive of one exchange step; returns a wait() callable.

    RCCL: one grouped call (ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd), device buffers, asynchronous on
    RCCL's stream -- wait() makes the current stream wait for it.  Other backends (gloo, used by the tests that run
    several ranks on one GPU): staged through host memory, same peers, same order."""
    if not sends and not recvs:
        return lambda: None
    if _is_nccl(group):
        p2p = [dist.P2POp(dist.isend, t, dist.get_global_rank(group, p) if group is not None else p, group) for t, p in sends]
        p2p += [dist.P2POp(dist.irecv, t, dist.get_global_rank(group, p) if group is not None else p, group) for t, p in recvs]
        works = dist.batch_isend_irecv(p2p)

        def wait():
            for w in works:
                w.wait()
        return wait
    grank = (lambda p: dist.get_global_rank(group, p)) if group is not None else (lambda p: p)
    host_in = [(t, torch.empty(t.shape, dtype=t.dtype), p) for t, p in recvs]
    works = [dist.isend(t.detach().cpu().contiguous(), grank(p), group=group) for t, p in sends]
    works += [dist.irecv(h, grank(p), group=group) for _, h, p in host_in]

    def wait():
        for w in works:
            w.wait()
        for t, h, _ in host_in:
            t.copy_(h)
    return wait


def _all_gather_rows(full, shard, group):
    """full (G*S, C) <- every rank's shard (S, C), rank-major."""
    if _is_nccl(group):
        dist.all_gather_into_tensor(full, shard, group=group)
    else:
        G = dist.get_world_size(group)
        S = shard.shape[0]
        dist.all_gather([full[r * S:(r + 1) * S] for r in range(G)], shard, group=group)


# --------------------------------------------------------------------------------------------- AANet across ranks

class HipAttentionOps(object):
    """The local pieces of the sharded AANet on the HIP kernels."""

    def scores(self, x_stack, scope):
        """x_stack (n,D,h,w,8) -> [S|R] (n,D,h,w,16): one launch over the local views."""
        st = variables.default_store()
        wu = st.get_host('%s/attention_activation/weight_unique' % scope, (3, 3, 3, 8, 8))
        ws = st.get_host('%s/attention_activation/weight_shared' % scope, (3, 3, 3, 8, 8))
        w16 = np.concatenate([ws, wu], axis=-1)
        return ops.conv(x_stack, scope + '/attention_activation/shared|unique', w16, relu=True, groups=x_stack.shape[0])

    def combine(self, srs, xs, out):
        """out (rows, 8) = sum_n softmax_n(R_n - S_n + sum_m S_m) X_n over row-slices srs[n] (rows,16), xs[n] (rows,8)."""
        return ops.aanet_combine(srs, xs, out=out)


def _attention_steps(x_stack, my_views, num_views, scope, shape, device, world, rank, impl=None):
    """Generator form of the voxel-transposed AANet of one group (see the module docstring).

    x_stack: (n_local, D,h,w,8) this rank's views (my_views, ascending) or None; shape = (D,h,w).  Yields
    ('comm', fn): fn(group) must be called eagerly by every rank of the group at that point (it posts / completes
    RCCL calls); returns the aggregated (D,h,w,8) volume, identical on every rank."""
    impl = impl or HipAttentionOps()
    D, h, w = shape
    V = D * h * w
    S = -(-V // world)
    lo = lambda r: min(r * S, V)                 # noqa: E731
    rows = lambda r: min((r + 1) * S, V) - lo(r)   # noqa: E731
    own = owner_of(num_views, world)
    views = sorted(own)
    mine = list(my_views)
    nloc = len(mine)
    f32 = dict(dtype=torch.float32, device=device)
    x_rows = x_stack.reshape(nloc, V, 8) if nloc else None
    # receive buffers for the other ranks' views (my row range)
    rx = {u: torch.empty((rows(rank), 8), **f32) for u in views if own[u] != rank}
    rsr = {u: torch.empty((rows(rank), 16), **f32) for u in views if own[u] != rank}
    state = {}

    def post_x(group):         # the X rows can travel while the 8->16 convolution runs
        sends = [(x_rows[i, lo(p):lo(p) + rows(p)], p) for i, v in enumerate(mine) for p in range(world)
                 if p != rank and rows(p) > 0]
        recvs = [(rx[u], own[u]) for u in views if own[u] != rank and rows(rank) > 0]
        state['wait_x'] = _p2p(sends, recvs, group)
    yield 'comm', post_x

    sr_rows = impl.scores(x_stack, scope).reshape(nloc, V, 16) if nloc else None

    def post_sr(group):
        sends = [(sr_rows[i, lo(p):lo(p) + rows(p)], p) for i, v in enumerate(mine) for p in range(world)
                 if p != rank and rows(p) > 0]
        recvs = [(rsr[u], own[u]) for u in views if own[u] != rank and rows(rank) > 0]
        wait_sr = _p2p(sends, recvs, group)
        state['wait_x']()
        wait_sr()
    yield 'comm', post_sr

    # combine over ALL views in view order on my rows
    shard = torch.empty((S, 8), **f32)
    if rows(rank) > 0:
        srs, xs = [], []
        for u in views:
            if own[u] == rank:
                i = mine.index(u)
                srs.append(sr_rows[i, lo(rank):lo(rank) + rows(rank)])
                xs.append(x_rows[i, lo(rank):lo(rank) + rows(rank)])
            else:
                srs.append(rsr[u])
                xs.append(rx[u])
        impl.combine(srs, xs, shard[:rows(rank)])
    full = torch.empty((world * S, 8), **f32)

    def gather(group):
        _all_gather_rows(full, shard, group)
    yield 'comm', gather
    return full[:V].reshape(D, h, w, 8)


def _drive(gen, group=None):
    """Run a step generator eagerly: perform every yielded communication step, return the generator's result."""
    try:
        while True:
            item = next(gen)
            if item[0] == 'comm':
                item[1](group)
            else:                      # ('allreduce', tensor, op)
                dist.all_reduce(item[1], op=item[2], group=group)
    except StopIteration as e:
        return e.value


def sharded_attention(local_xs, my_views, num_views, scope, like, impl=None, group=None):
    """AANet over views that live on different ranks of `group`.

    local_xs: (n_local, D,h,w,8) stack of this rank's views `my_views` (ascending source ids) or None; `like`: a
    tensor giving the (D,h,w) shape / device.  Returns the aggregated (D,h,w,8) volume, identical on every rank.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shape = tuple(like.shape[-4:-1])
    return _drive(_attention_steps(local_xs, my_views, num_views, scope, shape, like.device, world, rank, impl), group)


# --------------------------------------------------------------------------------------------- the pipeline of one rank

class HipLocalStages(object):
    """The rank-local compute of the sharded pipeline on the HIP kernels (model.*_batch: every network evaluated once over
    all the (view, direction) calls the rank owns).  The CPU tests substitute an oracle-backed object with the same four
    methods (tests/test_parallel_gloo.py); the product default is this class."""

    attention = None          # -> HipAttentionOps()

    def base(self, images, cams, max_d, depth_start, depth_interval, fwd, rev):
        """Base stage (reference model.py:398-417) of the owned directions -> (filtered cost volumes (len(fwd),D,h,w,8) in
        `fwd` order or None, {v: depth_view (h,w) for v in rev})."""
        from .atvsnet import model
        views = sorted(set(fwd) | set(rev))
        local = [0] + views                                    # the reference view and the owned sources
        index = {v: i for i, v in enumerate(local)}
        feats = model.feature_extraction_batch(torch.cat([images[:, i] for i in local], 0).unsqueeze(0))
        filtered, _, _, dview = model.base_stage_batch(feats, cams, max_d, depth_start, depth_interval, fwd=fwd, rev=rev,
                                                       ref_i=0, feature_index=index)
        h, w = images.shape[2] // 4, images.shape[3] // 4
        return (filtered if fwd else None), {v: dview[v].reshape(h, w) for v in rev}

    def head(self, cost_agg, max_d, depth_start, depth_interval):
        """AAM1's output conv + soft-argmin: (D,h,w,8) -> (prob_agg (1,D,h,w), depth_init (1,h,w,1))."""
        from .atvsnet import model
        prob_agg = model.output_conv(cost_agg.unsqueeze(0), reuse=False)
        return prob_agg, model.prob2depth(prob_agg, max_d, depth_start, depth_interval)

    def refine(self, images, cams, max_d, depth_start, depth_interval, fwd, depth_init, dviews, prob_agg, cost_agg):
        """Refinement of the owned sources in one pass -> refined cost volumes (len(fwd),D,h,w,8) = cost_agg + residual."""
        from .atvsnet import model
        local = [0] + list(fwd)
        index = {v: i for i, v in enumerate(local)}
        shallow = model.shallow_feature_batch(torch.cat([images[:, i] for i in local], 0).unsqueeze(0))
        cres, _ = model.refinement_batch(depth_init, dviews, prob_agg, cams, max_d, depth_start, depth_interval, list(fwd),
                                         shallow, ref_id=0, shallow_index=index)
        refined = torch.empty_like(cres)
        for b in range(len(fwd)):
            ops.add_n([cost_agg, cres[b]], out=refined[b])
        return refined

    def final(self, rcost_agg, max_d, depth_start, depth_interval):
        """AAM2's output conv, x4 upsample + soft-argmin: (D,h,w,8) -> (1,H,W,1)."""
        from .atvsnet import model
        rprob_agg = model.output_conv_refine(rcost_agg.unsqueeze(0), reuse=False)
        return model.prob2depth_upsample(rprob_agg, max_d, depth_start, depth_interval)[1]


def _sharded_steps(images, cams, max_d, world, rank, stages=None):
    """example.infer_multiview for this rank's share of the source views, as a generator: local compute runs
    between the yields, every yield is a communication step all ranks of the group must perform.
    Returns the full-resolution inverse-depth map (1,H,W,1), identical on every rank of the group.
    A rank may own no forward direction (more ranks than sources with split directions) or nothing at all (more ranks
    than directions): it still takes part in every exchange and ends with the same map."""
    from .atvsnet import example as ex
    stages = stages or HipLocalStages()
    n = images.shape[1]
    mine = plan(n, world)[rank]
    fwd = sorted(v for kind, v in mine if kind == 'fwd')
    rev = sorted(v for kind, v in mine if kind == 'rev')
    depth_start, depth_interval = ex.depth_range(cams)
    dev = images.device
    h, w = images.shape[2] // 4, images.shape[3] // 4
    dv_all = torch.zeros((n - 1, h, w), dtype=torch.float32, device=dev)

    # ---- base stage: every owned (view, direction) in ONE pass of each network
    filtered = None
    if fwd or rev:
        filtered, dview = stages.base(images, cams, max_d, depth_start, depth_interval, fwd, rev)
        for v in rev:      # quirk C11: the reverse direction swept the reference camera's depth range
            dv_all[v - 1].copy_(dview[v])
    yield 'allreduce', dv_all, dist.ReduceOp.SUM                # every view has exactly one contributor

    # ---- AAM1 across the group
    cost_agg = yield from _attention_steps(filtered, fwd, n, 'attention_aggregate', (max_d, h, w), dev, world, rank,
                                           stages.attention)
    prob_agg, depth_init = stages.head(cost_agg, max_d, depth_start, depth_interval)
    del filtered

    # ---- refinement of the owned sources, one pass
    refined = None
    if fwd:
        dviews = {v: dv_all[v - 1].reshape(1, h, w, 1) for v in fwd}
        refined = stages.refine(images, cams, max_d, depth_start, depth_interval, fwd, depth_init, dviews, prob_agg, cost_agg)

    # ---- AAM2 across the group, head, x4 upsample + soft-argmin (replicated)
    rcost_agg = yield from _attention_steps(refined, fwd, n, 'attention_aggregate_refine', (max_d, h, w), dev, world, rank,
                                            stages.attention)
    return stages.final(rcost_agg, max_d, depth_start, depth_interval)


def infer_multiview_sharded(images, cams, max_d=None, group=None, view_streams=True, stages=None):
    """example.infer_multiview with the source views sharded over the process group (every launch issued from
    Python).  Every rank returns the same full-resolution inverse-depth map (1,H,W,1).  (view_streams is accepted
    for compatibility: a rank evaluates its views in one batched pass.  stages: test hook, see HipLocalStages.)"""
    max_d = FLAGS.max_d if max_d is None else max_d
    return _drive(_sharded_steps(images, cams, max_d, dist.get_world_size(group), dist.get_rank(group), stages), group)


class ShardedGraphedInference(object):
    """The sharded pipeline as a chain of HIP graphs with the communication steps between them.

    The local compute between two communication steps is captured once into a HIP graph; the RCCL calls themselves stay
    ordinary calls on the tensors the graphs own (posted in the same order by every rank of the group).  One step =
    replay, communicate, replay, ... instead of hundreds of launches issued from Python per rank.  All graphs share
    one memory pool and are replayed in capture order.  Inputs live in static buffers: pass new images / cams to
    __call__."""

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
            _drive(_sharded_steps(self.images, self.cams, self.max_d, world, rank), group)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graphs, self.colls = [], []
        pool = torch.cuda.graph_pool_handle()
        gen = _sharded_steps(self.images, self.cams, self.max_d, world, rank)
        done = False
        while not done:
            g = torch.cuda.CUDAGraph()
            item = None
            # thread_local: the RCCL watchdog thread polls its events (hipEventQuery) while this thread captures;
            # in the default global mode that call is "not permitted when stream is capturing" and aborts the process
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter('always')
                with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                    try:
                        item = next(gen)
                    except StopIteration as e:
                        self.out, done = e.value, True
            # a segment that only allocates (the receive buffers between the depth-view all-reduce and the first AANet's first
            # exchange) captures NO node: torch says so in a warning at capture_end -- the exact signal -- and such a segment is
            # kept as None (its slot in the chain stays: graphs[i] is followed by colls[i]) instead of being replayed per map
            empty = any('Graph is empty' in str(w.message) for w in caught)
            for w in caught:
                if 'Graph is empty' not in str(w.message):
                    warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
            self.graphs.append(None if empty else g)
            if not done:
                # during capture nothing ran: the tensors hold whatever the pool had; performing the step keeps the
                # ranks' communication sequences aligned and is repeated, on real data, in every replay
                self._comm(item)
                self.colls.append(item)
        torch.cuda.synchronize(dev)
        self._weights = (ops.cache_snapshot(), variables.default_store().device_snapshot())

    def _comm(self, item):
        if item[0] == 'comm':
            item[1](self.group)
        else:
            dist.all_reduce(item[1], op=item[2], group=self.group)

    def __call__(self, images=None, cams=None):
        if images is not None:
            self.images.copy_(images)
        if cams is not None:
            self.cams.copy_(cams)
        for i, g in enumerate(self.graphs):
            if g is not None:
                g.replay()
            if i < len(self.colls):
                self._comm(self.colls[i])
        return self.out

    def timed_call(self):
        """One step with HIP events around every graph replay and every communication step (on the compute stream, which
        waits for RCCL's): {'graphs_ms': [...], 'comm_ms': [...]} of this rank."""
        ev = lambda: torch.cuda.Event(enable_timing=True)     # noqa: E731
        marks = [ev()]
        marks[0].record()
        kinds = []
        for i, g in enumerate(self.graphs):
            if g is not None:
                g.replay()
            m = ev()
            m.record()
            marks.append(m)
            kinds.append('graph')
            if i < len(self.colls):
                self._comm(self.colls[i])
                m = ev()
                m.record()
                marks.append(m)
                kinds.append('comm')
        torch.cuda.synchronize()
        ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(kinds))]
        g_ms = [round(t, 3) for t, k in zip(ms, kinds) if k == 'graph']
        c_ms = [round(t, 3) for t, k in zip(ms, kinds) if k == 'comm']
        return {'graphs_ms': g_ms, 'comm_ms': c_ms, 'graphs_total_ms': round(sum(g_ms), 3),
                'comm_total_ms': round(sum(c_ms), 3)}
