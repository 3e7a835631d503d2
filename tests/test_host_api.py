"""CPU: the host side of the drop-in boundary -- operator API plumbing (meta tensors: same host
code, no launches), variable naming, the C-ABI library (loads, exports every declared symbol,
host-side packing functions), refusal of CPU tensors (no fallback)."""
import ctypes

import numpy as np
import pytest
import torch

import atvsnet_amd                                   # noqa: F401
from atvsnet_amd import _lib, ops, variables
from atvsnet_amd.cnn_wrapper import atvsnet as nets
from atvsnet_amd.cnn_wrapper.network import Network
from oracle import tf_ops as T


def meta(*shape):
    return torch.empty(shape, dtype=torch.float32, device='meta')


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 30 and 'atvs_conv_mfma_f32' in names and 'atvs_build_cost_volume' in names
    for n in names:
        assert hasattr(L, n), n
    assert L.atvs_abi_version() >= 1 and L.atvs_target_arch() == b'gfx950'


def test_cpu_tensors_are_refused_no_fallback():
    x = torch.zeros(4, 5, 6)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.softargmin(x, torch.zeros(1), torch.zeros(1))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.conv(torch.zeros(4, 4, 4, 8), 'k', np.zeros((3, 3, 3, 8, 8), np.float32))
    with pytest.raises(TypeError):
        ops.add_n([meta(4).double(), meta(4).double()])


def test_network_plumbing_and_errors():
    class Tiny(Network):
        def setup(self):
            (self.feed('data').conv_bn(3, 8, 1, name='a').conv(3, 4, 2, name='b'))
            self.feed('a', 'a').add(name='c')
            self.feed('c').conv(1, 2, 1)          # auto-named

    net = Tiny({'data': meta(1, 8, 8, 8, 4)}, is_training=True)
    assert set(net.layers) == {'data', 'a', 'b', 'c', 'conv_1'}
    assert net.get_output() is net.layers['conv_1']
    assert net.get_output_by_name('b').shape == (1, 4, 4, 4, 4)
    assert net.get_shape_by_name('a') == (1, 8, 8, 8, 8)
    assert net.get_unique_name('conv') == 'conv_2'

    class NoInput(Network):
        def setup(self):
            self.conv(3, 8, 1, name='x')
    with pytest.raises(RuntimeError, match='No input variables found for layer x'):
        NoInput({'data': meta(1, 4, 4, 4, 4)}, is_training=True)

    class BadFeed(Network):
        def setup(self):
            self.feed('nope')
    with pytest.raises(KeyError, match='Unknown layer name fed: nope'):
        BadFeed({'data': meta(1, 4, 4, 4, 4)}, is_training=True)

    class BadRank(Network):
        def setup(self):
            self.feed('data').conv(3, 8, 1, name='r')
    with pytest.raises(ValueError, match='Improper input rank for layer: r'):
        BadRank({'data': meta(1, 4, 4)}, is_training=True)

    with pytest.raises(NotImplementedError):
        Network.setup(net)
    # layers off the hot path (reference network.py:218-268, 354-376, 411-508, 619-647, 657-689, 699-775) exist as
    # thin torch fallbacks with the reference's signatures
    class OffPath(Network):
        def setup(self):
            (self.feed('data').relu(name='r').max_pool(3, 2, name='mp').l2_pool(2, 2, name='lp')
                 .lrn(2, 1e-4, 0.75, name='n').sigmoid(name='s').nn_softmax(name='sm', axis=-1).l2norm(name='l2')
                 .reduce_mean([1, 2], name='gap').softmax(name='p').fc(5, name='fc', relu=False).dropout(name='do'))
            self.feed('r', 'r').multiply(name='m')
            self.feed('data').deconv(3, 6, 2, name='up').transpose([0, 3, 1, 2], name='t')
            self.feed('data').expand_dims(-1, name='e').tile([1, 1, 1, 1, 2], name='ti').squeeze(name='sq')
            self.feed('data').split_separable_conv2d(3, 7, 2, name='sep')
    o = OffPath({'data': meta(1, 8, 8, 4)}, is_training=False)
    assert o.get_output_by_name('mp').shape == (1, 4, 4, 4) and o.get_output_by_name('lp').shape == (1, 2, 2, 4)
    assert o.get_output_by_name('fc').shape == (1, 5) and o.get_output_by_name('up').shape == (1, 16, 16, 6)
    assert o.get_output_by_name('t').shape == (1, 6, 16, 16) and o.get_output_by_name('sep').shape == (1, 8, 8, 7)
    assert o.get_output_by_name('ti').shape == (1, 8, 8, 4, 2)
    # feeding another network's layer by [network, name] (reference network.py:98-104)
    class Borrow(Network):
        def setup(self):
            self.feed([net, 'a']).conv(1, 3, 1, name='z')
    assert Borrow({}, is_training=True).get_output().shape == (1, 8, 8, 8, 3)


@pytest.mark.parametrize('H,W,D', [(128, 160, 32), (512, 640, 192), (480, 928, 256)])
def test_shapes_of_every_network(H, W, D):
    h, w = H // 4, W // 4
    t = nets.ResNetDS2SPP({'data': meta(1, H, W, 3)}, is_training=True)
    assert t.get_output().shape == (1, h, w, 32)
    assert t.get_output_by_name('concat_feature').shape == (1, h, w, 320)
    assert t.get_output_by_name('branch_0_pool').shape == (1, -(-h // 64), -(-w // 64), 128)
    assert nets.ResNetDS2SPP_shallow_f16({'data': meta(1, H, W, 3)}, is_training=True).get_output().shape == (1, h, w, 16)
    u = nets.StackedUNet_prob({'data': meta(1, D, h, w, 64)}, is_training=True)
    assert u.get_output().shape == (1, D, h, w, 1)
    assert u.get_output_by_name('conv_b2_6_1').shape == (1, D, h, w, 8)
    assert u.get_output_by_name('conv_b1_3_1').shape == (1, D // 8, h // 8, w // 8, 64)
    assert 'conv_b2_6_2' not in nets.StackedUNet({'data': meta(1, D, h, w, 64)}, is_training=True).layers
    r = nets.CostVolRefineNet({'photo_group': meta(1, D, h, w, 48), 'geo_group': meta(1, D, h, w, 19),
                               'prob_vol': meta(1, D, h, w, 1), 'vis_hull': meta(1, D, h, w, 1)}, is_training=True)
    assert r.get_output().shape == (1, D, h, w, 1)
    assert r.get_output_by_name('global_refine_3dconv6_1').shape == (1, D, h, w, 8)
    a = nets.AttAggregation_keepchannel({'data': [meta(1, D, h, w, 8)] * 3}, is_training=True)
    assert a.get_output().shape == (1, D, h, w, 8)
    a6 = nets.AttAggregation({'data': meta(1, D, h, w, 8, 4)}, is_training=True)
    assert a6.get_output().shape == (1, D, h, w, 1)


def test_pipeline_dry_run_touches_exactly_the_variable_table():
    from atvsnet_amd.atvsnet import example as ex
    store = variables.default_store()
    saved = dict(store.host)
    store.clear()
    try:
        assert ex.infer_twoview(meta(1, 2, 128, 160, 3), meta(1, 2, 2, 4, 4), 32).shape == (1, 128, 160, 1)
        assert ex.infer_multiview(meta(1, 4, 128, 160, 3), meta(1, 4, 2, 4, 4), 32).shape == (1, 128, 160, 1)
        spec = dict(variables.variable_specs())
        assert set(store.host) == set(spec)
        assert all(tuple(store.host[k].shape) == tuple(spec[k]) for k in spec)
        assert sum(int(np.prod(s)) for s in spec.values()) == 3272059       # 3.27 M parameters
    finally:
        store.clear()
        store.host.update(saved)


def test_variable_store_is_deterministic_per_name_and_checks_shapes(tmp_path):
    a = variables.VariableStore().init_synthetic(7)
    b = variables.VariableStore()
    b.seed = 7
    k = b.get_host('conv_b0_0_1/conv3d/kernel', (3, 3, 3, 64, 8))
    assert np.array_equal(k, a.host['conv_b0_0_1/conv3d/kernel'])           # creation order does not matter
    assert abs(float(k.std()) - np.sqrt(2.0 / (27 * 72))) < 2e-3             # Xavier-normal
    with pytest.raises(ValueError, match='has shape'):
        b.get_host('conv_b0_0_1/conv3d/kernel', (3, 3, 3, 64, 16))
    path = str(tmp_path / 'w.npz')
    a.save_npz(path)
    c = variables.VariableStore()
    c.load_npz(path)
    assert set(c.host) == set(a.host)
    with pytest.raises(KeyError):
        c.get_host('not/a/variable', (1,))


def _emulate_gather_conv(x, packed, table, vec, J, NT, cout, out_shape, stride):
    """The K order of conv_mfma_f32_kernel, in numpy: group g = 4j+q reads `vec` channels at its tap."""
    Di, Hi, Wi, Cin = x.shape
    y = np.zeros(tuple(out_shape) + (cout,), np.float64)
    wp = packed.reshape(J, NT, 4, 16, vec)
    tab = table.reshape(J * 4, 4)
    for j in range(J):
        for q in range(4):
            dz, dy, dx, c0 = tab[j * 4 + q]
            if dz >= (1 << 27):
                assert not wp[j, :, q].any()
                continue
            wq = wp[j, :, q].reshape(NT * 16, vec)[:cout]                  # (cout, vec)
            for o in np.ndindex(*out_shape):
                z, yy, xx = o[0] * stride + dz, o[1] * stride + dy, o[2] * stride + dx
                if 0 <= z < Di and 0 <= yy < Hi and 0 <= xx < Wi:
                    y[o] += wq @ x[z, yy, xx, c0:c0 + vec]
    return y


@pytest.mark.parametrize('cin,cout,stride', [(8, 8, 1), (3, 16, 2), (20, 8, 1), (64, 32, 1)])
def test_conv_packing_host_function(cin, cout, stride):
    """atvs_conv_pack (host C function): packed weights + group table reproduce the convolution."""
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(1, 3, 4, 5, cin, generator=g)
    w = torch.randn(3, 3, 3, cin, cout, generator=g)
    want = T.conv(x, w, stride, 'SAME')[0].double().numpy()
    pads = [T.same_pad(s, 3, stride)[0] for s in (3, 4, 5)]
    taps = ops.conv_taps((3, 3, 3), 1, pads)
    pk = ops.pack_conv_weights(('t', cin, cout, stride), w.numpy(), taps, False, 'meta')
    L = _lib.lib()
    pf, ti = ctypes.c_long(), ctypes.c_long()
    L.atvs_conv_pack_size(27, cin, cout, None, None, None, ctypes.byref(pf), ctypes.byref(ti))
    packed, table = np.empty(pf.value, np.float32), np.empty(ti.value, np.int32)
    tp = np.ascontiguousarray(np.array(taps, np.int32))
    wn = np.ascontiguousarray(w.numpy())
    assert L.atvs_conv_pack(wn.ctypes.data_as(ctypes.c_void_p), 0, tp.ctypes.data_as(ctypes.c_void_p), 27, cin, cout,
                            packed.ctypes.data_as(ctypes.c_void_p), table.ctypes.data_as(ctypes.c_void_p)) == 0
    assert pk.vec == (4 if cin % 4 == 0 else 1) and pk.ntiles * 16 >= cout
    got = _emulate_gather_conv(x[0].double().numpy(), packed.astype(np.float64), table, pk.vec, pk.ksteps, pk.ntiles,
                               cout, want.shape[:3], stride)
    assert np.abs(got - want).max() < 1e-4
    assert L.atvs_conv_pack_size(27, cin, 129, None, None, None, None, None) == -2


def test_deconv_parity_classes_cover_every_tap_once():
    seen = []
    for par in [(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]:
        taps = ops.deconv_s2_class_taps(par)
        assert len(taps) == 2 ** (3 - sum(par))
        seen += [t[0] for t in taps]
    assert sorted(seen) == list(range(27))
    assert ops.same_pad(128, 3, 2) == (0, 64) and ops.same_pad(128, 3, 1) == (1, 128) and ops.same_pad(7, 3, 2, 1) == (1, 4)


def test_flags_defaults_match_the_reference():
    from atvsnet_amd import FLAGS
    FLAGS.reset()
    assert (FLAGS.view_num, FLAGS.max_d, FLAGS.batch_size, FLAGS.inverse_depth, FLAGS.sample_scale) == (5, 128, 1, True, 0.25)
    assert FLAGS.example_index == 2 and FLAGS.num_gpus == 1


def test_library_abi_version_matches_header():
    from atvsnet_amd import _lib
    assert _lib.lib().atvs_abi_version() == _lib.header_abi_version() >= 2


def test_weight_reload_drops_arranged_copies():
    """The pack caches are keyed by variable NAME: setting a variable must drop every arranged copy."""
    import numpy as np
    import torch
    from atvsnet_amd import ops, variables
    store = variables.default_store()
    w = store.get_host('conv_b1_1_1/conv3d/kernel', (3, 3, 3, 16, 16)).copy()
    x = torch.empty((8, 16, 24, 16), dtype=torch.float32, device='meta')
    ops.conv(x, 'conv_b1_1_1/conv3d/kernel', w)
    ops._fold_cache[('k', ())] = 1
    ops._xp_cache['k'] = 1
    ops._virt_cache['k'] = 1
    assert ops._pack_cache
    store.set('conv_b1_1_1/conv3d/kernel', np.zeros_like(w))
    assert not ops._pack_cache and not ops._fold_cache and not ops._xp_cache and not ops._virt_cache
    store.set('conv_b1_1_1/conv3d/kernel', w)


def test_results_outside_the_fp16_range_fail_loudly():
    """The split-operand kernels turn an activation beyond +-65504 into inf/NaN; the host drivers refuse such a result by name
    instead of writing it to disk (atvsnet/example.py check_finite)."""
    from atvsnet_amd.atvsnet import example
    ok = np.linspace(0.001, 0.01, 12, dtype=np.float32).reshape(3, 4)
    assert example.check_finite(ok) is ok
    bad = ok.copy()
    bad[1, 2] = np.nan
    bad[2, 0] = np.inf
    with pytest.raises(FloatingPointError, match='2 non-finite.*ATVS_SPLIT16=0'):
        example.check_finite(bad)


def test_config_object_and_context_manager():
    """ops.cfg holds every dispatch switch; ops.configure sets them for a block and restores them -- also on an exception;
    unknown names raise; split16 drags the x-pair kernel choice along unless xb is given in the same call."""
    from atvsnet_amd import ops
    before = ops.cfg.snapshot()
    with ops.configure(split16=False, prologue=False):
        assert not ops.cfg.split16 and not ops.cfg.xb and not ops.cfg.prologue
        with ops.configure(split16=True, xb=False):
            assert ops.cfg.split16 and not ops.cfg.xb
        assert not ops.cfg.split16 and not ops.cfg.xb
    assert ops.cfg.snapshot() == before
    with pytest.raises(ZeroDivisionError):
        with ops.configure(bottleneck=False, split_off=('c2b', 'btl')):
            assert ops.cfg.split_off == frozenset(('c2b', 'btl')) and not ops.split_on('c2b') and ops.split_on('c16b')
            1 / 0
    assert ops.cfg.snapshot() == before
    with pytest.raises(AttributeError):
        ops.configure(no_such_switch=True)
    with pytest.raises(AttributeError):
        ops.cfg.no_such_switch = True
    with pytest.raises(ValueError):
        ops.cfg.force_impl = 'fastest'
    assert sum(1 for ln in open(ops.__file__) if ln.startswith('def use_')) == 0


def test_deferred_sums_on_meta_tensors():
    """A deferred skip sum (2 or 3 terms) reaches the transposed convolution unformed where the kernel takes it (16 -> 8) and is
    materialised everywhere else; the x-pair siblings only take two-term sums.  Host logic only (meta tensors: no launch)."""
    import torch
    from atvsnet_amd import ops
    G, shape = 2, (4, 8, 16)
    mk = lambda c: torch.empty((G,) + shape + (c,), dtype=torch.float32, device='meta')      # noqa: E731
    par = lambda c: torch.empty((G, 3, c), dtype=torch.float32, device='meta')               # noqa: E731
    three = ops.PendingSum([ops.PendingBN(mk(16), par(16), True), mk(16), ops.PendingBN(mk(16), par(16), False)])
    assert ops.deconv_sum_ok(three, 8, G) and not ops.deconv_sum_ok(three, 16, G)
    assert not ops.siblings_prologue_ok(ops.PendingSum([ops.PendingBN(mk(8), par(8), True)] * 3))
    with pytest.raises(ValueError):
        ops.PendingSum([mk(16)])
    with pytest.raises(ValueError):
        three.prologue()
    with ops.configure(sum_on_load=False):
        assert not ops.deconv_sum_ok(three, 8, G)
    w = np.zeros((3, 3, 3, 8, 16), np.float32)
    y, st = ops.conv3d_transpose_s2(three, ('meta-up', 16), w, want_stats=True, groups=G)
    assert tuple(y.shape) == (G, 8, 16, 32, 8) and three._final is None and st.groups == G
    # a planar pending batch norm enters a sum materialised (its raw buffer is not a channel-last operand)
    D, h, w_ = 4, 8, 32
    planar = ops.PendingBN(torch.empty((1, 1, ops.planar_stride(D, h, w_)), dtype=torch.float32, device='meta'),
                           torch.empty((3, 8), dtype=torch.float32, device='meta'), True, planar=(D, h, w_))
    s2 = ops.PendingSum([planar, torch.empty((1, D, h, w_, 8), dtype=torch.float32, device='meta')])
    assert all(not isinstance(t, ops.PendingBN) for t in s2.items)
