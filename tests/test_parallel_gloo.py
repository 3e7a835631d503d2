"""CPU: the view-sharded multi-GPU path with world_size 2 on the gloo backend.

The exchange of a-tvsnet_amd/parallel.py (all-to-all of voxel shards of [S|R|X], combine on the shard, all-gather) is
run by two / three processes; the local pieces (8->16 convolution, the cross-view softmax on a row range) are
supplied by the CPU oracle through the `impl` hook (tests only -- the product default is the HIP kernels), and the
result must equal the single-process oracle AANet over all views."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleAttentionOps(object):
    """The local pieces of the sharded AANet supplied by the CPU oracle (interface of parallel.HipAttentionOps)."""

    def __init__(self, W):
        self.W = W

    def scores(self, x_stack, scope):
        from oracle import tf_ops as T
        ws = self.W['%s/attention_activation/weight_shared' % scope]
        wu = self.W['%s/attention_activation/weight_unique' % scope]
        out = []
        for x in x_stack:
            s = torch.clamp(T.conv(x[None], ws, 1, 'SAME'), min=0)[0]
            r = torch.clamp(T.conv(x[None], wu, 1, 'SAME'), min=0)[0]
            out.append(torch.cat([s, r], -1))
        return torch.stack(out)

    def combine(self, srs, xs, out):
        S = [t[..., :8] for t in srs]
        R = [t[..., 8:] for t in srs]
        ssum = sum(S[1:], S[0])
        U = torch.stack([(R[i] - S[i]) + ssum for i in range(len(srs))], -1)
        p = torch.softmax(U, dim=-1)
        out.copy_((p * torch.stack(list(xs), -1)).sum(dim=-1))      # the oracle's own form (oracle/nets.py attention_aggregation)
        return out


class OracleLocalStages(object):
    """parallel.HipLocalStages' four methods on the CPU oracle (tests only): lets the COMPLETE sharded pipeline -- plan,
    depth_view all-reduce, both AANet exchanges, ranks without a forward view -- run on gloo without a GPU."""

    def __init__(self, W):
        self.W = W
        self.attention = OracleAttentionOps(W)

    def base(self, images, cams, max_d, ds, di, fwd, rev):
        from oracle import model as OM, nets
        W = self.W
        ref_f = nets.resnet_ds2_spp(images[:, 0], W)
        feats = {v: nets.resnet_ds2_spp(images[:, v], W) for v in sorted(set(fwd) | set(rev))}
        filt, dview = [], {}
        for v in fwd:
            cv = OM.build_cost_volume(ref_f, feats[v], cams, max_d, ds, di, 0, v)
            filt.append(OM.cost_volume_reasoning(cv, W)[1][0])
        for v in rev:
            cv = OM.build_cost_volume(feats[v], ref_f, cams, max_d, ds, di, v, 0)
            dview[v] = OM.prob2depth(OM.cost_volume_reasoning(cv, W)[0], max_d, ds, di)[0, ..., 0]
        return (torch.stack(filt) if filt else None), dview

    def head(self, cost_agg, max_d, ds, di):
        from oracle import model as OM, nets
        prob_agg = nets.output_conv(cost_agg[None], self.W, 'attention_prob_vol')
        return prob_agg, OM.prob2depth(prob_agg, max_d, ds, di)

    def refine(self, images, cams, max_d, ds, di, fwd, depth_init, dviews, prob_agg, cost_agg):
        from oracle import model as OM
        return torch.stack([OM.TVSNet_refine(depth_init, dviews[v], prob_agg, cost_agg[None], images, cams, max_d, ds, di,
                                             self.W, view_i=v, ref_i=0)[1][0] for v in fwd])

    def final(self, rcost_agg, max_d, ds, di):
        from oracle import model as OM, nets
        rprob = nets.output_conv(rcost_agg[None], self.W, 'attention_prob_vol_refine')
        return OM.prob2depth_upsample(rprob, max_d, ds, di)[1]


def _pipeline_worker(rank, world, port, n_views, size, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import atvsnet_amd                                 # noqa: F401
        from atvsnet_amd import parallel, synthetic, variables
        store = variables.VariableStore().init_synthetic(1234)
        W = {k: torch.from_numpy(v) for k, v in store.host.items()}
        H, Wd, D = size
        imgs, cams = synthetic.make_inputs(n_views, H, Wd, D)
        with torch.no_grad():
            out = parallel.infer_multiview_sharded(torch.from_numpy(imgs), torch.from_numpy(cams), D, stages=OracleLocalStages(W))
        q.put((rank, parallel.plan(n_views, world)[rank], out.numpy().copy()))
    except Exception as e:                                   # fail the test at once instead of letting the parent time out
        q.put((rank, 'error', repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n_views', [(8, 9), (4, 3), (3, 5), (5, 3)])
def test_sharded_pipeline_equals_single_process(weights, world, n_views):
    """The complete view-sharded depth-map pipeline (a-tvsnet_amd/parallel.py: plan, base stage per owner, depth_view
    all-reduce, AAM1 exchange, refinement per owner, AAM2 exchange, head) over gloo against the single-process oracle
    pipeline.  (8, 9) = BASELINE configs[3]'s partition: 8 sources, one per rank; (4, 3) = two ranks per source, the odd
    ranks own only a reverse direction (no forward view, nothing to refine); (3, 5) = uneven shares; (5, 3) = one rank
    owns nothing at all and still takes part in every exchange."""
    import numpy as np
    from atvsnet_amd import synthetic
    from oracle import model as OM
    size = (64, 96, 32)                                      # features 16 x 24, D = 32: every U-Net level divides
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + world * 17 + n_views
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, n_views, size, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in range(world):
        item = q.get(timeout=600)
        if item[1] == 'error':
            for p in procs:
                p.kill()                                     # the exact children started above
            pytest.fail('rank %d: %s' % (item[0], item[2]))
        res.append(item)
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    imgs, cams = synthetic.make_inputs(n_views, *size)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)          # as the workers: oneDNN's blocking (and so its rounding) depends on the thread count
    try:
        with torch.no_grad():
            want = OM.run_multiview(torch.from_numpy(imgs), torch.from_numpy(cams), weights, size[2]).numpy()
    finally:
        torch.set_num_threads(threads)
    for r in res[1:]:
        assert np.array_equal(r[2], res[0][2])               # every rank ends with the same map
    rel = float(np.mean(np.abs(res[0][2] - want) / np.abs(want)))
    assert res[0][2].shape == want.shape and rel <= 1e-4, rel
    owners = [sorted(k for k, _ in r[1]) for r in res]
    if (world, n_views) == (4, 3):
        assert owners == [['fwd'], ['rev'], ['fwd'], ['rev']]
    if (world, n_views) == (5, 3):
        assert owners[4] == []
    if (world, n_views) == (8, 9):
        assert all(o == ['fwd', 'rev'] for o in owners)


def _worker(rank, world, port, nviews, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import atvsnet_amd                                 # noqa: F401
        from atvsnet_amd import parallel, variables
        store = variables.VariableStore().init_synthetic(1234)
        W = {k: torch.from_numpy(v) for k, v in store.host.items()}
        g = torch.Generator().manual_seed(5)
        X = torch.randn(nviews, 5, 7, 9, 8, generator=g)        # 315 voxels: uneven row shards
        mine = parallel.local_views(nviews + 1, rank, world)
        stack = torch.stack([X[v - 1] for v in mine]) if mine else None
        out = parallel.sharded_attention(stack, mine, nviews + 1, 'attention_aggregate', X[0], impl=OracleAttentionOps(W))
        q.put((rank, out.numpy().copy()))     # plain bytes: the worker may exit before the parent reads
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('nviews,world', [(4, 2), (1, 2), (3, 2), (4, 3), (8, 4)])
def test_sharded_attention_equals_single_process(weights, nviews, world):
    from oracle import nets
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + nviews * 5 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, nviews, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {r: torch.from_numpy(a) for r, a in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    X = torch.randn(nviews, 5, 7, 9, 8, generator=g)        # 315 voxels: uneven row shards
    want = nets.attention_aggregation(X.permute(1, 2, 3, 4, 0)[None].contiguous(), weights, 'attention_aggregate')[0]
    for r in range(1, world):
        assert torch.equal(outs[0], outs[r])                 # every rank ends with the same volume
    assert float((outs[0] - want).abs().max()) < 1e-5


def test_view_partition():
    from atvsnet_amd import parallel
    for n, world in ((5, 1), (5, 2), (5, 4), (5, 8), (9, 8), (3, 2), (3, 8), (5, 6)):
        tasks = parallel.plan(n, world)
        assert len(tasks) == world
        for kind in ('fwd', 'rev'):                           # every direction of every source exactly once
            assert sorted(v for t in tasks for (k, v) in t if k == kind) == list(range(1, n))
        owned = [parallel.local_views(n, r, world) for r in range(world)]
        assert sorted(v for o in owned for v in o) == list(range(1, n))
        if world >= 2 * (n - 1):                              # directions of a pair on two ranks
            assert all(len(t) <= 1 for t in tasks)
        else:
            busy = [len(o) for o in owned if o]
            assert max(busy) - min(busy) <= 1
