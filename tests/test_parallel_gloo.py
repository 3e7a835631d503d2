"""CPU: the view-sharded multi-GPU path with world_size 2 on the gloo backend.

The collective composition of a-tvsnet_amd/parallel.py (three all-reduces inside the AANet) is
run by two processes; the local pieces (8->16 convolution, partial reductions) are supplied by
the CPU oracle through the `impl` hook (tests only -- the product default is the HIP kernels),
and the result must equal the single-process oracle AANet over all views."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleAttentionOps(object):
    def __init__(self, W):
        self.W = W

    def scores(self, xs, scope):
        from oracle import tf_ops as T
        ws = self.W['%s/attention_activation/weight_shared' % scope]
        wu = self.W['%s/attention_activation/weight_unique' % scope]
        out = []
        for x in xs:
            s = torch.clamp(T.conv(x[None], ws, 1, 'SAME'), min=0)[0]
            r = torch.clamp(T.conv(x[None], wu, 1, 'SAME'), min=0)[0]
            out.append(torch.cat([s, r], -1))
        return out

    def partial(self, srs, xs, stage, ssum=None, umax=None):
        S = [t[..., :8] for t in srs]
        R = [t[..., 8:] for t in srs]
        if stage == 0:
            return sum(S[1:], S[0]).clone()      # fresh buffer: all_reduce works in place
        U = [(R[i] - S[i]) + ssum for i in range(len(srs))]
        if stage == 1:
            return torch.stack(U).max(0).values.contiguous()
        e = [torch.exp(u - umax) for u in U]
        return torch.stack([sum(e[1:], e[0]), sum([e[i] * xs[i] for i in range(1, len(xs))], e[0] * xs[0])])

    def divide(self, num, den):
        return num / den


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
        X = torch.randn(nviews, 6, 8, 10, 8, generator=g)
        mine = [X[v - 1] for v in parallel.local_views(nviews + 1, rank, world)]
        out = parallel.sharded_attention(mine, 'attention_aggregate', X[0], impl=OracleAttentionOps(W))
        q.put((rank, out.numpy().copy()))     # plain bytes: the worker may exit before the parent reads
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('nviews', [4, 1, 3])
def test_sharded_attention_equals_single_process(weights, nviews):
    from oracle import nets
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + nviews
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nviews, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = {r: torch.from_numpy(a) for r, a in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    X = torch.randn(nviews, 6, 8, 10, 8, generator=g)
    want = nets.attention_aggregation(X.permute(1, 2, 3, 4, 0)[None].contiguous(), weights, 'attention_aggregate')[0]
    assert torch.equal(outs[0], outs[1])                     # every rank ends with the same volume
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
