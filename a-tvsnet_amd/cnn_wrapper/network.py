"""Operator API of the reference's ``cnn_wrapper/network.py`` on the HIP kernels.

Same class, method names, keyword arguments, chaining (`feed(...).op(...).op(...)`),
name-based layer lookup and error behaviour as /root/reference/cnn_wrapper/network.py
(cited per method).  Differences that follow from leaving TensorFlow-1.5 graph mode:

* execution is eager: ``setup()`` runs the layers on the device as it builds them;
* tensors are batch-first, channel-last float32 torch tensors (device memory only; every op on the
  hot path is a gfx950 kernel behind ``include/atvsnet_hip.h`` reached through ``ops``); ``meta``
  tensors run the same host code without launching anything (shape / memory planning);
* variables live in ``variables.default_store()`` under the TF variable names and are
  always shared by name (the reference passes reuse=tf.AUTO_REUSE or builds once);
* the reference evaluates every network with batch 1 (FLAGS.batch_size) and training-mode batch norm,
  i.e. statistics per CALL (quirk C1).  A batch B > 1 is accepted only with
  ``independent_samples=True`` and then means "B calls of this network at once": every launch
  covers the B samples, the batch-norm statistics stay per sample, and the results equal B
  separate calls.  (TensorFlow would pool the statistics over such a batch; the reference never
  builds one.)
"""
import numpy as np
import torch

from .. import ops
from .. import variables

DEFAULT_PADDING = 'SAME'
BN_EPS = 1e-3      # tf.layers.batch_normalization / slim.batch_norm default epsilon

string_types = (str,)


def layer(op):
    """Decorator for composable network layers (reference network.py:11-34)."""

    def layer_decorated(self, *args, **kwargs):
        # Automatically set a name if not provided.
        name = kwargs.setdefault('name', self.get_unique_name(op.__name__))
        # Figure out the layer inputs.
        if not self.terminals:
            raise RuntimeError('No input variables found for layer %s.' % name)
        elif len(self.terminals) == 1:
            layer_input = self.terminals[0]
        else:
            layer_input = list(self.terminals)
        layer_output = op(self, layer_input, *args, **kwargs)
        self.layers[name] = layer_output
        self.feed(layer_output)
        return self

    layer_decorated.__name__ = op.__name__
    layer_decorated.__doc__ = op.__doc__
    return layer_decorated


class Network(object):
    """Class NetWork (reference network.py:37-139)."""

    def __init__(self, inputs, is_training, dropout_rate=0.9, seed=None, reuse=False, scope_name=None,
                 independent_samples=False):
        self.inputs = inputs
        self.terminals = []
        self.layers = dict(inputs)
        self.trainable = is_training
        self.reuse = reuse
        self.scope_name = scope_name
        self.training = is_training
        self.seed = seed
        self.dropout_rate = dropout_rate
        self.independent_samples = bool(independent_samples)
        self.store = variables.default_store()
        self.setup()

    def setup(self):
        '''Construct the network. '''
        raise NotImplementedError('Must be implemented by the subclass.')

    def load(self, data_path, session=None, ignore_missing=False):
        '''Load network weights from a numpy-serialized {op_name: {param_name: array}} dict
        (reference network.py:67-82); `session` is accepted and ignored.'''
        data_dict = np.load(data_path, allow_pickle=True).item()
        for op_name in data_dict:
            for param_name, data in data_dict[op_name].items():
                self.store.set('%s/%s' % (op_name, param_name), data)

    def feed(self, *args):
        '''Set the input(s) for the next operation by replacing the terminal nodes.
        The arguments can be either layer names or the actual layers (reference :84-107).'''
        assert args
        self.terminals = []
        for fed_layer in args:
            if isinstance(fed_layer, string_types):
                try:
                    fed_layer = self.layers[fed_layer]
                except KeyError:
                    raise KeyError('Unknown layer name fed: %s' % fed_layer)
            elif isinstance(fed_layer, list):
                if len(fed_layer) == 2 and isinstance(fed_layer[0], Network) and isinstance(fed_layer[1], string_types):
                    try:
                        fed_layer = fed_layer[0].get_output_by_name(fed_layer[1])
                    except KeyError:
                        raise KeyError('Unknown layer name fed: %s' % fed_layer[1])
            self.terminals.append(fed_layer)
        return self

    def get_output(self):
        '''Returns the current network output.'''
        t = self.terminals[-1]
        return t.materialize() if isinstance(t, ops.LAZY) else t

    def get_output_by_name(self, layer_name):
        t = self.layers[layer_name]
        return t.materialize() if isinstance(t, ops.LAZY) else t

    def get_shape_by_name(self, layer_name):
        '''Shape of a layer (a tuple here; tf.shape in the reference :121-127).
        A list of per-view tensors (attention input) reports the stacked shape.'''
        t = self.layers[layer_name]
        if isinstance(t, (list, tuple)):
            return tuple(t[0].shape) + (len(t),)
        return tuple(t.shape)

    def get_unique_name(self, prefix):
        '''Returns an index-suffixed unique name for the given prefix (reference :129-134).'''
        ident = sum(t.startswith(prefix) for t, _ in self.layers.items()) + 1
        return '%s_%d' % (prefix, ident)

    def change_inputs(self, inputs):
        assert len(inputs) == 1
        for key in inputs:
            self.layers[key] = inputs[key]

    def make_var(self, name, shape, initializer=None, device=None):
        '''A variable by (scoped) name (reference :275-279): created on first use.'''
        return self.store.get(name, shape, device if device is not None else 'meta')

    def validate_padding(self, padding):
        assert padding in ('SAME', 'VALID')

    # ------------------------------------------------------------------ helpers
    def _bt(self, x, what):
        """The dense batch-first tensor of a layer input; B > 1 only for independent samples."""
        if isinstance(x, (ops.SplitVolume,) + ops.LAZY):
            x = x.materialize()
        if x.shape[0] != 1 and not self.independent_samples:
            raise ValueError('%s: batch size must be 1 (FLAGS.batch_size), got %d; pass independent_samples=True to '
                             'evaluate several calls of the network at once' % (what, x.shape[0]))
        return x if x.is_contiguous() else x.contiguous()      # e.g. a concat-buffer slice consumed on its own

    def _kernel(self, name, shape):
        """Host copy of a kernel variable (the pack cache uploads the arranged form)."""
        return self.store.get_host(name, shape)

    def _vec(self, name, n, like):
        return self.store.get(name, (n,), like.device)

    def _bn(self, y, st, scope, center, relu, C=None, c_off=0):
        """tf.layers.batch_normalization(center, scale=False, training=self.training) [+ relu].
        C / c_off: y is a wider concat buffer and only that channel slice is this layer's output."""
        Cn = y.shape[-1] if C is None else C
        beta = self._vec('%s/batch_normalization/beta' % scope, Cn, y) if center else None
        if self.training and C is not None:
            # slice of a concat buffer: only the moments now; one dense normalisation pass over the whole
            # buffer happens in concat() (a strided pass per slice would touch 32-byte fragments of each row)
            self._pending_bn = getattr(self, '_pending_bn', {})
            self._pending_bn.setdefault(id(y), []).append((c_off, C, ops.bn_params(st, C, y, beta, BN_EPS), relu))
            return y
        if self.training:
            return ops.batch_norm(y, st, beta=beta, relu=relu, inplace=True, eps=BN_EPS, C=C, c_off=c_off,
                                  groups=y.shape[0])
        if C is not None:
            raise NotImplementedError('moving-average batch norm into a concat slice')
        return self._bn_inference(y, '%s/batch_normalization' % scope, beta, relu)

    def concat_buffer(self, name, like, channels):
        """Pre-allocate the output of a later `concat(name=...)` so that the layers feeding it can write their
        channel slice directly (conv_bn(..., out_slice=(name, c_off))) instead of being copied."""
        shape = tuple(like.shape[:-1]) + (channels,)
        buf = torch.empty(shape, dtype=torch.float32, device=like.device)
        self._concat_bufs = getattr(self, '_concat_bufs', {})
        self._concat_bufs[name] = buf
        return buf

    def _bn_inference(self, y, scope, beta, relu):
        """Moving-average BN (never used by the reference's inference code, which passes is_training=True)."""
        C = y.shape[-1]
        mean = self.store.get_host('%s/moving_mean' % scope, (C,))
        var = self.store.get_host('%s/moving_variance' % scope, (C,))
        b = self.store.get_host(scope + '/beta', (C,)) if beta is not None else np.zeros(C, np.float32)
        p = np.stack([mean, 1.0 / np.sqrt(var + BN_EPS), b]).astype(np.float32)
        params = torch.empty((3, C), dtype=torch.float32, device='meta') if y.is_meta else torch.from_numpy(p).to(y.device)
        return ops.bn_apply(y, params, relu)

    # ------------------------------------------------------------------ layers on the hot path
    @layer
    def conv(self, input, kernel_size, filters, strides, name, relu=True, padding=DEFAULT_PADDING, biased=False,
             rate=1):
        '''tf.layers.conv2d / conv3d (reference network.py:141-169): variables name/kernel [, name/bias].'''
        rank = input.dim()
        if rank not in (4, 5):
            raise ValueError('Improper input rank for layer: ' + name)
        if rank == 4 and padding == 'SAME' and ops.norm_on_load_2d_ok(input, kernel_size, filters, strides, rate) \
                and (input.shape[0] == 1 or self.independent_samples):
            # a pending batch norm (+ ReLU) applied while the kernel stages its input: the normalised tensor is never written
            raw, (_, params, _, in_relu, _) = input.prologue()
            w = self._kernel('%s/kernel' % name, (kernel_size,) * 2 + (raw.shape[-1], filters))
            bias = self._vec('%s/bias' % name, filters, raw) if biased else None
            return ops.conv(raw, name + '/kernel', w, stride=1, dilation=rate, padding='SAME', bias=bias, relu=relu,
                            groups=raw.shape[0], in_params=params, in_relu=in_relu)
        x = self._bt(input, name)
        G, cin = x.shape[0], x.shape[-1]
        if rank == 5 and kernel_size == 3 and filters == 1 and cin == 8 and strides == 1 and rate == 1 \
                and padding == 'SAME' and not biased and not relu:
            # the 8 -> 1 probability heads: HBM-bound, dedicated FMA kernel
            wd = self.store.get('%s/kernel' % name, (3, 3, 3, 8, 1), x.device)
            return ops.conv3d_8to1(x, wd, groups=G)
        w = self._kernel('%s/kernel' % name, (kernel_size,) * (rank - 2) + (cin, filters))
        bias = self._vec('%s/bias' % name, filters, x) if biased else None
        return ops.conv(x, name + '/kernel', w, stride=strides, dilation=rate, padding=padding, bias=bias, relu=relu,
                        groups=G)

    @layer
    def conv_bn(self, input, kernel_size, filters, strides, name, relu=True, center=False, padding=DEFAULT_PADDING,
                biased=False, rate=1, out_slice=None, defer_bn=False):
        '''conv (no activation) -> batch norm with batch statistics -> relu (reference network.py:172-215):
        variables name/conv{2,3}d/kernel.  The statistics come from the convolution's epilogue.'''
        rank = input.dim()
        if rank not in (4, 5):
            raise ValueError('Improper input rank for layer: ' + name)
        buf, c_off = None, 0
        if out_slice is not None:          # extension: write into a channel slice of a pre-allocated concat buffer
            buf, c_off = self._concat_bufs[out_slice[0]], int(out_slice[1])
        if isinstance(input, ops.SplitVolume):
            if kernel_size == 3 and rate == 1 and padding == 'SAME' and not biased and self.training:
                if input.shape[0] != 1 and not self.independent_samples:
                    raise ValueError('%s: batch size must be 1 (FLAGS.batch_size)' % name)
                vname = '%s/conv3d/kernel' % name
                w = self._kernel(vname, (3, 3, 3, input.shape[-1], filters))
                y, st = ops.conv_split(input, vname, w, stride=strides, want_stats=True, out=buf, y_coff=c_off)
                if defer_bn and buf is None and not center and filters % 4 == 0:
                    return ops.PendingBN(y, ops.bn_params(st, filters, y, None, BN_EPS), relu)
                return self._slice_out(self._bn(y, st, name, center, relu, C=(filters if buf is not None else None),
                                                c_off=c_off), out_slice, filters)
            input = input.materialize()
        in_params, in_relu, in_sum = None, False, None
        if rank == 4 and padding == 'SAME' and self.training and buf is None \
                and ops.norm_on_load_2d_ok(input, kernel_size, filters, strides, rate) \
                and (input.shape[0] == 1 or self.independent_samples):
            # the producer's batch norm (+ ReLU) is applied while this convolution stages its input
            x, (_, in_params, _, in_relu, _) = input.prologue()
        elif rank == 5 and padding == 'SAME' and self.training and buf is None and isinstance(input, ops.LAZY) \
                and ops.norm_on_load_3d_ok(input, kernel_size, filters, strides, rate) \
                and (input.shape[0] == 1 or self.independent_samples):
            # 3-D: a pending batch norm, or the U-Net's skip sum of two, formed while the halo is staged
            x, pro = input.prologue()
            if pro is not None and isinstance(input, ops.PendingSum):
                x1, in_params, p1, in_relu, r1 = pro
                in_sum = (x1, p1, r1)
            elif pro is not None:
                in_params, in_relu = pro[1], pro[3]
        else:
            x = self._bt(input, name)
        G, cin = x.shape[0], x.shape[-1]
        kind = 'conv2d' if rank == 4 else 'conv3d'
        vname = '%s/%s/kernel' % (name, kind)
        w = self._kernel(vname, (kernel_size,) * (rank - 2) + (cin, filters))
        bias = self._vec('%s/%s/bias' % (name, kind), filters, x) if biased else None
        if self.training and defer_bn and buf is None and not center and filters % 4 == 0:
            # extension: the layer's consumers are adds only -> hand them the raw output + the moments
            y, st = ops.conv(x, vname, w, stride=strides, dilation=rate, padding=padding, bias=bias, want_stats=True,
                             groups=G, in_params=in_params, in_relu=in_relu, in_sum=in_sum)
            return ops.PendingBN(y, ops.bn_params(st, filters, y, None, BN_EPS), relu)
        if self.training:
            y, st = ops.conv(x, vname, w, stride=strides, dilation=rate, padding=padding, bias=bias, want_stats=True,
                             out=buf, y_coff=c_off, groups=G, in_params=in_params, in_relu=in_relu, in_sum=in_sum)
        else:
            y, st = ops.conv(x, vname, w, stride=strides, dilation=rate, padding=padding, bias=bias, out=buf,
                             y_coff=c_off, groups=G), None
        return self._slice_out(self._bn(y, st, name, center, relu, C=(filters if buf is not None else None), c_off=c_off),
                               out_slice, filters)

    def conv_bn_siblings(self, first, second):
        """Extension: two conv_bn layers of the SAME input (the current terminal), e.g. the U-Net's conv_b*_0_1
        (8 channels, stride 1) and its encoder branch conv_b*_1_0 (16 channels, stride 2) -- reference
        cnn_wrapper/atvsnet.py feeds one tensor to both.  `first` / `second` are the keyword arguments of the two
        conv_bn calls (kernel_size, filters, strides, name [, relu, defer_bn]).  When the kernels allow it both run
        as ONE launch that stages the input once; otherwise this is exactly the two conv_bn calls.  Both outputs
        are registered under their names; the terminal becomes `second` (so the encoder chain can continue)."""
        if len(self.terminals) != 1:
            raise RuntimeError('conv_bn_siblings takes one input')
        src = self.terminals[0]
        a, b = dict(first), dict(second)
        plain = lambda d: (d.get('kernel_size', 3) == 3 and not d.get('center', False) and not d.get('biased', False)   # noqa: E731
                           and d.get('padding', DEFAULT_PADDING) == 'SAME' and d.get('rate', 1) == 1
                           and d.get('out_slice') is None)
        shape = tuple(src.shape)
        fusable = (self.training and len(shape) == 5 and (shape[0] == 1 or self.independent_samples) and plain(a)
                   and plain(b) and a['strides'] == 1 and b['strides'] == 2)
        if fusable and isinstance(src, ops.LAZY) and not ops.siblings_prologue_ok(src):
            src = src.materialize()        # a lazy input the kernel cannot form on load
        if fusable:
            if isinstance(src, ops.SplitVolume):
                cin_var = src.cv
                fusable = ops.siblings_ok(shape[1:4], cin_var, a['filters'], b['filters'])
            else:
                fusable = ops.siblings_ok(shape[1:4], shape[4], a['filters'], b['filters'])
        if not fusable:
            if isinstance(src, ops.LAZY):
                src = src.materialize()
            self.feed(src).conv_bn(**a)
            self.feed(src).conv_bn(**b)
            return self
        cin = shape[4]
        va, vb = '%s/conv3d/kernel' % a['name'], '%s/conv3d/kernel' % b['name']
        wa = self._kernel(va, (3, 3, 3, cin, a['filters']))
        wb = self._kernel(vb, (3, 3, 3, cin, b['filters']))
        if isinstance(src, ops.SplitVolume):
            (ya, sa), (yb, sb) = ops.conv_split_siblings(src, va, wa, vb, wb)
        else:
            # a lazy src (pending batch norm / pending sum) is formed inside the launch: normalise- and add-on-load
            xin = src if isinstance(src, ops.LAZY) else self._bt(src, a['name'])
            (ya, sa), (yb, sb) = ops.conv_siblings(xin, va, wa, vb, wb, groups=shape[0])
        if a.get('defer_bn', False):
            out_a = ops.PendingBN(ya, ops.bn_params(sa, a['filters'], ya, None, BN_EPS), a.get('relu', True))
        else:
            out_a = self._bn(ya, sa, a['name'], False, a.get('relu', True))
        if b.get('defer_bn', False):
            out_b = ops.PendingBN(yb, ops.bn_params(sb, b['filters'], yb, None, BN_EPS), b.get('relu', True))
        else:
            out_b = self._bn(yb, sb, b['name'], False, b.get('relu', True))
        self.layers[a['name']] = out_a
        self.layers[b['name']] = out_b
        self.feed(out_b)
        return self

    def refine_stems(self, concat_name, stems, filters=8):
        """Extension: the four input stems of CostVolRefineNet (reference cnn_wrapper/atvsnet.py:300-313) and their concat,
        i.e. exactly  feed(src).conv_bn(3, 8, 1, name=stem) for (src, stem) in `stems`; feed(*stems).concat(-1, concat_name).
        When the inputs have the form the refinement builds (photo / geo as SplitVolumes with 16 and 2 D-varying
        channels, 1-channel probability and visual-hull volumes) the geo | prob | vishull stems run as ONE HBM-bound pass
        that stores whole rows of the 32-channel concat (with the raw photo-stem output passed through), and one dense
        batch-norm + ReLU pass follows; otherwise the stems are issued one by one.  Registers every stem and the concat
        under their names; the terminal becomes the concat."""
        (s_photo, n_photo), (s_geo, n_geo), (s_prob, n_prob), (s_hull, n_hull) = stems
        photo, geo, prob, hull = (self.layers[k] for k in (s_photo, s_geo, s_prob, s_hull))
        fused = (self.training and filters == 8 and ops.cfg.stem and ops.cfg.force_impl is None
                 and isinstance(photo, ops.SplitVolume) and isinstance(geo, ops.SplitVolume)
                 and geo.var.shape[-1] == 2 and not isinstance(prob, (ops.SplitVolume, ops.PendingBN))
                 and not isinstance(hull, (ops.SplitVolume, ops.PendingBN)) and prob.dim() == 5 and hull.dim() == 5
                 and prob.shape[-1] == 1 and hull.shape[-1] == 1
                 and (photo.shape[0] == 1 or self.independent_samples))
        if not fused:
            self.concat_buffer(concat_name, prob, 4 * filters)
            names = []
            for i, (src, name) in enumerate(stems):
                self.feed(src).conv_bn(3, filters, 1, name=name, out_slice=(concat_name, i * filters))
                names.append(name)
            return self.feed(*names).concat(axis=-1, name=concat_name)
        B = photo.shape[0]
        vp, vg = '%s/conv3d/kernel' % n_photo, '%s/conv3d/kernel' % n_geo
        wp = self._kernel(vp, (3, 3, 3, photo.shape[-1], filters))
        wg = self._kernel(vg, (3, 3, 3, geo.shape[-1], filters))
        wpr = self._kernel('%s/conv3d/kernel' % n_prob, (3, 3, 3, 1, filters))
        wh = self._kernel('%s/conv3d/kernel' % n_hull, (3, 3, 3, 1, filters))
        wg_var, planes_g = ops._fold_split_weights(vg, wg, geo.chan_map, geo.var.shape[-1], geo.const.shape[-1])
        pb_geo = ops.conv(geo.const, (vg, 'planes'), planes_g, groups=B)                   # (B,h,w,24)
        dhw = tuple(int(v) for v in photo.shape[1:4])
        planar = dhw if (ops.planar_concat_ok(dhw) and (not photo.planar or photo.pieces) and photo.cv % 8 == 0) else None
        if planar:
            # the concat as four dense 8-channel planes: the photo stem writes plane 0, the FMA stems planes 1..3, and the
            # consumer (3dconv0_1 | 3dconv1_0) stages 32-byte voxels per chunk instead of 32 of every 128 bytes
            buf = torch.empty((B, 4, ops.planar_stride(*dhw)), dtype=torch.float32, device=pb_geo.device)
            st_photo = ops.conv_split_into_plane(photo, vp, wp, buf, 0, planar)
            buf, st24 = ops.refine_stems(None, geo.var, pb_geo, self._bt(prob, n_prob), self._bt(hull, n_hull),
                                         (vg, 'stems'), wg_var, wpr, wh, planar_out=buf)
        else:
            y_photo, st_photo = ops.conv_split(photo, vp, wp, want_stats=True)             # dense raw (B,D,h,w,8)
            buf, st24 = ops.refine_stems(y_photo, geo.var, pb_geo, self._bt(prob, n_prob), self._bt(hull, n_hull),
                                         (vg, 'stems'), wg_var, wpr, wh)
        pshape = (3, 4 * filters) if B == 1 else (B, 3, 4 * filters)
        params = torch.empty(pshape, dtype=torch.float32, device=buf.device)
        ops.copy_channels(ops.bn_params(st_photo, filters, buf, None, BN_EPS), params, filters, 0, 0)
        ops.copy_channels(ops.bn_params(st24, 3 * filters, buf, None, BN_EPS), params, 3 * filters, 0, filters)
        # the batch norm + ReLU of the 32 channels stays pending: the concat's consumer (3dconv0_1 | 3dconv1_0) normalises
        # on load; anything else that asks for a stem or the concat gets the materialised tensor
        out = ops.PendingBN(buf, params, True, planar=planar)
        for i, name in enumerate((n_photo, n_geo, n_prob, n_hull)):
            self.layers[name] = ops.LazySlice(out, i * filters, (i + 1) * filters)
        self.layers[concat_name] = out
        return self.feed(out)

    def _slice_out(self, y, out_slice, filters):
        """Layer result: the dense tensor, or (for out_slice) a view of the concat buffer tagged for concat()."""
        if out_slice is None:
            return y
        full = self._concat_bufs[out_slice[0]]
        view = full[..., int(out_slice[1]):int(out_slice[1]) + filters]
        view._atvs_slice = (out_slice[0], int(out_slice[1]))
        return view

    @layer
    def deconv_bn(self, input, kernel_size, filters, strides, name, relu=True, center=False, padding=DEFAULT_PADDING,
                  biased=False, defer_bn=False):
        '''conv3d_transpose -> batch norm -> relu (reference network.py:510-550): variables
        name/conv3d_transpose/kernel [k,k,k,Cout,Cin].  The path only uses k=3, stride 2, SAME, rank 5.'''
        rank = input.dim()
        if rank not in (4, 5):
            raise ValueError('Improper input rank for layer: ' + name + ', input_shape: ' + str(rank))
        if rank != 5 or kernel_size != 3 or strides != 2 or padding != 'SAME' or biased:
            raise NotImplementedError('deconv_bn: only the 3-D k=3, stride-2, SAME, unbiased form used by '
                                      'cnn_wrapper/atvsnet.py is built')
        if isinstance(input, ops.PendingSum) and self.training and (input.shape[0] == 1 or self.independent_samples):
            x = input                          # a deferred skip sum: formed inside the launch where the kernel can (ops.deconv_sum_ok)
        else:
            x = self._bt(input, name)
        G = x.shape[0]
        vname = '%s/conv3d_transpose/kernel' % name
        w = self._kernel(vname, (3, 3, 3, filters, x.shape[-1]))
        if self.training and defer_bn and not center and filters % 4 == 0:
            y, st = ops.conv3d_transpose_s2(x, vname, w, want_stats=True, groups=G)
            return ops.PendingBN(y, ops.bn_params(st, filters, y, None, BN_EPS), relu)
        if self.training:
            y, st = ops.conv3d_transpose_s2(x, vname, w, want_stats=True, groups=G)
        else:
            y, st = ops.conv3d_transpose_s2(x, vname, w, groups=G), None
        return self._bn(y, st, name, center, relu)

    def bottleneck(self, inputs, kernel_size, depth, stride=1, rate=1, name=None):
        """Bottleneck residual unit variant with BN before convolutions (reference network.py:552-602).

        slim.batch_norm (beta, batch statistics) + relu -> [1x1 shortcut on the pre-activation] ->
        1x1 (bias, relu) -> 3x3 (dilated | strided with explicit symmetric pad) (bias, relu) ->
        1x1 (bias) + shortcut (added in the last convolution's epilogue).
        """
        scope = name
        x = self._bt(inputs, scope)
        G = x.shape[0]
        depth_in = x.shape[-1]
        beta = self._vec('%s/preact/beta' % scope, depth_in, x)
        # pre-activation (slim.batch_norm + relu, :570-571).  Its consumers are 1x1 convolutions: where the GEMM kernel
        # serves them the normalisation is applied as they load x and the pre-activated tensor is never written.
        pre_params = None
        _pre = []

        def preact():
            if not _pre:
                if self.training:
                    _pre.append(ops.bn_apply(x, pre_params, True, out=torch.empty_like(x)))
                else:
                    _pre.append(self._bn_inference(x.clone(), '%s/preact' % scope, beta, True))
            return _pre[0]
        if self.training:
            # statistics of `inputs` come from the epilogue of the convolution that produced it, when
            # that was the previous bottleneck's conv3 (+shortcut); otherwise from a channel_stats pass
            st_in = getattr(inputs, '_atvs_stats', None)
            if st_in is None:
                st_in = ops.channel_stats(x, groups=G)
            pre_params = ops.bn_params(st_in, depth_in, x, beta, BN_EPS)
        if self.training and depth == depth_in and stride == 1 and kernel_size == 3 \
                and ops.bottleneck_ok(depth, rate, x.shape[1], x.shape[2]):
            # identity-shortcut unit: nothing global between the pre-activation's moments and the output -> ONE launch
            # (bottleneck_b.hip); r1 and r2 never leave the CU
            out, st = ops.bottleneck(
                x, pre_params, tuple(scope + '/%s/weights' % c for c in ('conv1', 'conv2', 'conv3')),
                self._kernel('%s/conv1/weights' % scope, (1, 1, depth, depth)), self._vec('%s/conv1/biases' % scope, depth, x),
                self._kernel('%s/conv2/weights' % scope, (3, 3, depth, depth)), self._vec('%s/conv2/biases' % scope, depth, x),
                self._kernel('%s/conv3/weights' % scope, (1, 1, depth, depth)), self._vec('%s/conv3/biases' % scope, depth, x),
                dilation=rate)
            out._atvs_stats = st
            return out
        on_load = self.training and ops.conv1x1_ok(depth_in, depth)
        if depth == depth_in:
            if stride == 1:
                shortcut = x
            else:
                raise NotImplementedError('bottleneck: max_pool shortcut (same depth, stride != 1) is never '
                                          'reached by cnn_wrapper/atvsnet.py and is not built')
        elif on_load and stride == 1:
            shortcut = ops.conv(x, scope + '/shortcut/weights',
                                self._kernel('%s/shortcut/weights' % scope, (1, 1, depth_in, depth)),
                                bias=self._vec('%s/shortcut/biases' % scope, depth, x), groups=G, in_params=pre_params,
                                in_relu=True)
        else:
            shortcut = ops.conv(preact(), scope + '/shortcut/weights',
                                self._kernel('%s/shortcut/weights' % scope, (1, 1, depth_in, depth)), stride=stride,
                                bias=self._vec('%s/shortcut/biases' % scope, depth, x), groups=G)
        w1 = self._kernel('%s/conv1/weights' % scope, (1, 1, depth_in, depth))
        b1 = self._vec('%s/conv1/biases' % scope, depth, x)
        if on_load:
            r = ops.conv(x, scope + '/conv1/weights', w1, bias=b1, relu=True, groups=G, in_params=pre_params, in_relu=True)
        else:
            r = ops.conv(preact(), scope + '/conv1/weights', w1, bias=b1, relu=True, groups=G)
        w2 = self._kernel('%s/conv2/weights' % scope, (kernel_size, kernel_size, depth, depth))
        b2 = self._vec('%s/conv2/biases' % scope, depth, x)
        if stride == 1 and self.training and kernel_size == 3 and ops.conv2d_tail_ok(depth, rate, r.shape[1], r.shape[2]):
            # conv2 and conv3 in one launch (the 128-channel dilated units: their conv1 halo does not fit the fully fused unit)
            out, st = ops.conv2d_tail(r, (scope + '/conv2/weights', scope + '/conv3/weights'), w2, b2,
                                      self._kernel('%s/conv3/weights' % scope, (1, 1, depth, depth)),
                                      self._vec('%s/conv3/biases' % scope, depth, x), residual=shortcut, dilation=rate)
            out._atvs_stats = st
            return out
        if stride == 1:
            r = ops.conv(r, scope + '/conv2/weights', w2, dilation=rate, bias=b2, relu=True, groups=G)
        else:
            k_eff = kernel_size + (kernel_size - 1) * (rate - 1)
            pb = (k_eff - 1) // 2
            pe = (k_eff - 1) - pb
            r = ops.conv(r, scope + '/conv2/weights', w2, stride=stride, dilation=rate,
                         explicit_pad=[(pb, pe), (pb, pe)], bias=b2, relu=True, groups=G)
        out, st = ops.conv(r, scope + '/conv3/weights', self._kernel('%s/conv3/weights' % scope, (1, 1, depth, depth)),
                           bias=self._vec('%s/conv3/biases' % scope, depth, x), residual=shortcut, want_stats=True,
                           groups=G)
        out._atvs_stats = st          # consumed by the next bottleneck's pre-activation batch norm
        return out

    @layer
    def res_block(self, inputs, kernel_size, depth, num_block=1, stride=1, rate=1, name=None):
        '''num_block bottlenecks, scopes name_0 .. name_{n-2}, name (reference network.py:604-616).'''
        if num_block == 1:
            return self.bottleneck(inputs=inputs, kernel_size=kernel_size, depth=depth, stride=stride, rate=rate,
                                   name=name)
        output = self.bottleneck(inputs=inputs, kernel_size=kernel_size, depth=depth, stride=stride, rate=rate,
                                 name=name + '_' + str(0))
        for i in range(1, num_block):
            scope_name = name + '_' + str(i) if i != num_block - 1 else name
            output = self.bottleneck(inputs=output, kernel_size=kernel_size, depth=depth, stride=1, rate=rate,
                                     name=scope_name)
        return output

    @layer
    def avg_pool(self, input, pool_size, strides, name, padding=DEFAULT_PADDING):
        '''tf.layers.average_pooling2d (reference network.py:665-671); SAME only.'''
        if padding != 'SAME':
            raise NotImplementedError('avg_pool: only SAME padding is built')
        x = self._bt(input, name)
        return ops.avg_pool_same(x, pool_size, strides, groups=x.shape[0])

    @layer
    def image_resize(self, input, size, name, align_corners=True, method='bilinear', out_slice=None):
        '''tf.image.resize_images: always bilinear (reference network.py:649-655, quirk C14).  out_slice=(concat name, channel
        offset) (extension): the result is written into that slice of a pre-allocated concat buffer (concat_buffer).'''
        if not align_corners:
            raise NotImplementedError('image_resize: only align_corners=True is built')
        x = self._bt(input, name)
        if out_slice is not None:
            buf, c_off = self._concat_bufs[out_slice[0]], int(out_slice[1])
            ops.resize_bilinear(x, (int(size[0]), int(size[1])), out=buf, c_off=c_off, groups=x.shape[0])
            return self._slice_out(buf, out_slice, x.shape[-1])
        return ops.resize_bilinear(x, (int(size[0]), int(size[1])), groups=x.shape[0])

    @layer
    def concat(self, inputs, axis, name):
        '''tf.concat (reference network.py:691-693); channel axis only.'''
        if axis not in (-1, inputs[0].dim() - 1):
            raise NotImplementedError('concat: only the channel axis is built')
        # inputs already written into their slices of a pre-allocated buffer (concat_buffer): nothing to copy
        tags = [getattr(t, '_atvs_slice', None) for t in inputs]
        if all(tg is not None and tg[0] == name for tg in tags):
            off = 0
            ok = True
            for t, tg in zip(inputs, tags):
                ok = ok and tg[1] == off
                off += t.shape[-1]
            buf = self._concat_bufs[name]
            if ok and off == buf.shape[-1]:
                pend = getattr(self, '_pending_bn', {}).pop(id(buf), [])
                if pend:
                    G = buf.shape[0]
                    pshape = (3, off) if G == 1 else (G, 3, off)
                    params = torch.empty(pshape, dtype=torch.float32, device=buf.device)
                    relus = set()
                    for c_off, C, p, relu in pend:
                        ops.copy_channels(p, params, C, 0, c_off)
                        relus.add(bool(relu))
                    assert len(relus) == 1 and sum(pc[1] for pc in pend) == off
                    ops.bn_apply(buf, params, relus.pop())
                return buf
        buf = getattr(self, '_concat_bufs', {}).get(name)
        if buf is not None and sum(t.shape[-1] for t in inputs) == buf.shape[-1] \
                and not getattr(self, '_pending_bn', {}).get(id(buf)):
            # some inputs were written in place: copy the others into their slices
            off = 0
            for t, tg in zip(inputs, tags):
                if not (tg is not None and tg[0] == name and tg[1] == off):
                    ops.copy_channels(self._bt(t, name), buf, t.shape[-1], 0, off)
                off += t.shape[-1]
            return buf
        return ops.concat_channels([self._bt(t, name) for t in inputs])

    @layer
    def add(self, inputs, name, defer=False, plus=None):
        '''tf.add_n (reference network.py:695-697).  Inputs whose batch norm is still pending (conv_bn /
        deconv_bn with defer_bn=True) are normalised inside the add kernel.  defer=True (extension): a sum of two or three
        whose consumer can add on load (conv_bn_siblings, deconv_bn) is handed over unformed.  plus (extension): the name of an
        input layer holding ONE sample; the layer `name + '_plus'` = that sample + this sum (per sample of the sum) is formed in
        the same pass where the add kernel runs, else by add_n per sample.'''
        inputs = [t.materialize() if isinstance(t, (ops.PendingSum, ops.LazySlice)) else t for t in inputs]
        if defer and plus is None and self.training and len(inputs) in (2, 3) and all(t.dim() == 5 for t in inputs):
            return ops.PendingSum([t if isinstance(t, ops.PendingBN) else self._bt(t, name) for t in inputs])
        base = None
        if plus is not None:
            base = self.layers[plus]
            base = (base[0] if base.dim() == inputs[0].dim() else base).contiguous()
        if len(inputs) in (2, 3) and any(isinstance(t, ops.PendingBN) for t in inputs) \
                and inputs[0].shape[-1] % 4 == 0:
            items = [t if isinstance(t, ops.PendingBN) else self._bt(t, name) for t in inputs]
            if base is None:
                return ops.bn_add(items)
            y, self.layers[name + '_plus'] = ops.bn_add(items, plus=base)
            return y
        y = ops.add_n([self._bt(t, name) for t in inputs])
        if base is not None:
            y2 = torch.empty_like(y)
            for b in range(y.shape[0]):
                ops.add_n([base, y[b]], out=y2[b])
            self.layers[name + '_plus'] = y2
        return y

    def attention_activation(self, input, kernel_size, name, filters=None, second_weight=False, relu=True,
                             padding=DEFAULT_PADDING, biased=False, n_view=None):
        raise NotImplementedError('attention_activation is fused into attention_aggregation on this backend')

    @layer
    def attention_aggregation(self, input, kernel_size, name, filters=None, second_weight=False, relu=True,
                              padding=DEFAULT_PADDING, biased=False, n_view=None):
        '''AANet aggregation over views (reference network.py:378-408 -> :282-351).

        input: (B,D,H,W,C,N) like the reference; or a list of N tensors (B,D,H,W,C) (no stacking copy); or ONE
        5-D tensor (N,D,H,W,C) = the N views stacked on the leading axis (B = 1; what the batched per-view
        networks produce).  Variables name/attention_activation/{weight_unique,weight_shared}.  The shared and
        unique 3x3x3 convolutions run as one C->2C convolution (one launch over all views when they are stacked);
        the cross-view softmax and the weighted sum are one kernel.  Only the form the path uses is built.
        '''
        if not (second_weight and relu and not biased and padding == 'SAME' and kernel_size == 3):
            raise NotImplementedError('attention_aggregation: only second_weight=True, relu=True, biased=False, '
                                      'kernel 3, SAME (cnn_wrapper/atvsnet.py:202,234) is built')
        stacked = None
        if isinstance(input, (list, tuple)):
            xs = [ops_b1(t, name) for t in input]
        elif input.dim() == 5:
            stacked = input if input.is_contiguous() else input.contiguous()
            xs = [stacked[n] for n in range(stacked.shape[0])]
        else:
            if input.dim() != 6:
                raise ValueError('Improper input rank for layer: ' + name)
            st = ops_b1(input, name)
            nv = st.shape[-1]
            xs = []
            for n in range(nv):
                xn = torch.empty(tuple(st.shape[:-1]), dtype=torch.float32, device=st.device)
                ops.copy_channels(st, xn.reshape(-1, 1), 1, n, 0)
                xs.append(xn)
        c_in = xs[0].shape[-1]
        if c_in != 8 or (filters not in (None, c_in)):
            raise NotImplementedError('attention_aggregation: C = filters = 8 is built')
        scope = '%s/attention_activation' % name
        wu = self.store.get_host('%s/weight_unique' % scope, (3, 3, 3, c_in, c_in))
        ws = self.store.get_host('%s/weight_shared' % scope, (3, 3, 3, c_in, c_in))
        key = scope + '/shared|unique'
        if ops.aanet_fused_ok(xs):
            # the whole module in one launch: [S|R] of every view stays in registers (aanet_b.hip)
            return ops.aanet_fused(xs, key, ws, wu).unsqueeze(0)
        w16 = np.concatenate([ws, wu], axis=-1)
        if stacked is not None:
            sr = ops.conv(stacked, key, w16, relu=True, groups=stacked.shape[0])
            srs = [sr[n] for n in range(sr.shape[0])]
        else:
            srs = [ops.conv(x, key, w16, relu=True) for x in xs]
        return ops.aanet_combine(srs, xs).unsqueeze(0)

    # ------------------------------------------------------------------ API surface off the hot path
    # The remaining reference layers (network.py:218-268, 354-376, 411-508, 619-647, 657-689, 699-775) are never
    # used by cnn_wrapper/atvsnet.py.  They are provided as thin fallbacks on torch's device operators (MIOpen /
    # ATen), i.e. NOT on the hand-written kernels and outside every parity / performance claim of this repository.
    @layer
    def relu(self, input, name):
        return torch.relu(input)

    @layer
    def transpose(self, input, perm, name, conjugate=False):
        return input.permute(*perm).contiguous()

    @layer
    def divide(self, input, denominator, name):
        return input / denominator

    @layer
    def reduce_mean(self, input, axis, name, keepdims=True):
        return input.mean(dim=axis, keepdim=keepdims)

    @layer
    def reduce_sum(self, input, axis, name, keepdims=True):
        return input.sum(dim=axis, keepdim=keepdims)

    @layer
    def tile(self, input, multiples, name):
        return input.repeat(*multiples)

    @layer
    def squeeze_and_transpose(self, input, squeeze_dims, perm, name, conjugate=False):
        x = input
        for d in sorted(squeeze_dims, reverse=True):
            x = x.squeeze(d)
        return x.permute(*perm).contiguous()

    def _pool_nchw(self, x, pool_size, strides, padding, fn):
        """SAME / VALID window pooling of a (B,H,W,C) tensor through torch (fn = max | avg with valid counts)."""
        import torch.nn.functional as F
        x = x.permute(0, 3, 1, 2)
        if padding == 'SAME':
            H, W = x.shape[2:]
            ph = max((-(-H // strides) - 1) * strides + pool_size - H, 0)
            pw = max((-(-W // strides) - 1) * strides + pool_size - W, 0)
            pad = (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)
            if fn == 'max':
                x = F.pad(x, pad, value=float('-inf'))
                y = F.max_pool2d(x, pool_size, strides)
            else:
                ones = F.pad(torch.ones_like(x[:, :1]), pad)
                y = F.avg_pool2d(F.pad(x, pad), pool_size, strides) / F.avg_pool2d(ones, pool_size, strides)
        else:
            y = F.max_pool2d(x, pool_size, strides) if fn == 'max' else F.avg_pool2d(x, pool_size, strides)
        return y.permute(0, 2, 3, 1).contiguous()

    @layer
    def max_pool(self, input, pool_size, strides, name, padding=DEFAULT_PADDING):
        return self._pool_nchw(input, pool_size, strides, padding, 'max')

    @layer
    def l2_pool(self, input, pool_size, strides, name, padding=DEFAULT_PADDING):
        return torch.sqrt(self._pool_nchw(input * input, pool_size, strides, padding, 'avg'))

    @layer
    def lrn(self, input, radius, alpha, beta, name, bias=1.0):
        # tf.nn.local_response_normalization: x / (bias + alpha * sum_{|j-i| <= radius} x_j^2) ** beta
        import torch.nn.functional as F
        sq = (input * input).unsqueeze(1)
        k = 2 * radius + 1
        s = F.avg_pool3d(F.pad(sq, (radius, radius)), (1, 1, k), stride=1) * k if input.dim() == 4 else None
        if s is None:
            raise ValueError('Improper input rank for layer: ' + name)
        return input / torch.pow(bias + alpha * s.squeeze(1), beta)

    @layer
    def multiply(self, inputs, name):
        return inputs[0] * inputs[1]

    @layer
    def multiply_channel_wise(self, inputs, name):
        return inputs[0] * inputs[1]        # broadcasting = tile(inputs[0]) to the channels of inputs[1]

    @layer
    def fc(self, input, num_out, name, relu=True):
        n_in = input.shape[-1]
        w = self.store.get('%s/kernel' % name, (n_in, num_out), input.device)
        b = self.store.get('%s/bias' % name, (num_out,), input.device)
        y = input @ w + b
        return torch.relu(y) if relu else y

    @layer
    def sigmoid(self, input, name):
        return torch.sigmoid(input)

    @layer
    def softmax(self, input, name, dim=-1):
        if input.dim() > 2:
            if input.shape[1] == 1 and input.shape[2] == 1:
                input = input.squeeze(2).squeeze(1)
            else:
                raise ValueError('Rank 2 tensor input expected for softmax!')
        return torch.softmax(input, dim=dim)

    @layer
    def nn_softmax(self, input, name, axis=-1):
        return torch.softmax(input, dim=axis)

    @layer
    def batch_normalization(self, input, name, center=True, scale=True, relu=False):
        '''tf.layers.batch_normalization with its default training=False: moving averages, gamma (the reference
        forces scale=True), optional beta.'''
        C = input.shape[-1]
        dev = input.device
        mean = self.store.get('%s/moving_mean' % name, (C,), dev)
        var = self.store.get('%s/moving_variance' % name, (C,), dev)
        y = (input - mean) * torch.rsqrt(var + BN_EPS) * self.store.get('%s/gamma' % name, (C,), dev)
        if center:
            y = y + self.store.get('%s/beta' % name, (C,), dev)
        return torch.relu(y) if relu else y

    @layer
    def dropout(self, input, name):
        '''slim.dropout(keep_prob=self.dropout_rate, is_training=self.training).'''
        if not self.training:
            return input
        g = torch.Generator(device=input.device)
        g.manual_seed(0 if self.seed is None else int(self.seed))
        keep = float(self.dropout_rate)
        mask = (torch.rand(input.shape, generator=g, device=input.device) < keep).to(input.dtype)
        return input * mask / keep

    @layer
    def l2norm(self, input, name, dim=-1):
        return input * torch.rsqrt(torch.clamp((input * input).sum(dim=dim, keepdim=True), min=1e-12))

    @layer
    def deconv(self, input, kernel_size, filters, strides, name, relu=True, padding=DEFAULT_PADDING, biased=False):
        '''tf.layers.conv2d_transpose / conv3d_transpose (reference network.py:479-508), SAME, kernel [k.., Cout, Cin].'''
        import torch.nn.functional as F
        rank = input.dim()
        if rank not in (4, 5):
            raise ValueError('Improper input rank for layer: ' + name)
        if padding != 'SAME':
            raise NotImplementedError('deconv: SAME padding')
        nsp = rank - 2
        cin = input.shape[-1]
        kname = '%s/kernel' % name
        w = self.store.get(kname, (kernel_size,) * nsp + (filters, cin), input.device)
        x = input.permute(0, rank - 1, *range(1, rank - 1))
        wt = w.permute(nsp + 1, nsp, *range(nsp))                     # (Cin, Cout, k..)
        f = F.conv_transpose2d if nsp == 2 else F.conv_transpose3d
        y = f(x, wt, stride=strides)
        # SAME: output = input * stride; crop the (k - stride) surplus, begin-light like the gradient of a SAME conv
        sl = [slice(None), slice(None)]
        for a in range(nsp):
            tot = kernel_size - strides
            b0 = max(tot, 0) // 2
            sl.append(slice(b0, b0 + input.shape[1 + a] * strides))
        y = y[tuple(sl)].permute(0, *range(2, rank), 1).contiguous()
        if biased:
            y = y + self.store.get('%s/bias' % name, (filters,), input.device)
        return torch.relu(y) if relu else y

    @layer
    def split_separable_conv2d(self, inputs, kernel_size, filters, rate, name, weight_decay=0.00004,
                               depthwise_weights_initializer_stddev=0.33, pointwise_weights_initializer_stddev=0.06):
        '''slim.separable_conv2d (depthwise, ReLU) then slim.conv2d 1x1 (ReLU), reference network.py:218-268.'''
        import torch.nn.functional as F
        C = inputs.shape[-1]
        dev = inputs.device
        wd = self.store.get('%s_depthwise/depthwise_weights' % name, (kernel_size, kernel_size, C, 1), dev)
        bd = self.store.get('%s_depthwise/biases' % name, (C,), dev)
        wp = self.store.get('%s_pointwise/weights' % name, (1, 1, C, filters), dev)
        bp = self.store.get('%s_pointwise/biases' % name, (filters,), dev)
        x = inputs.permute(0, 3, 1, 2)
        pad = rate * (kernel_size - 1) // 2
        y = F.conv2d(x, wd.permute(2, 3, 0, 1), bd, padding=pad, dilation=rate, groups=C)
        y = torch.relu(y)
        y = torch.relu(F.conv2d(y, wp.permute(3, 2, 0, 1), bp))
        return y.permute(0, 2, 3, 1).contiguous()

    @layer
    def attention_activation_layer(self, input, kernel_size, name, filters=None, second_weight=False, relu=True,
                                   padding=DEFAULT_PADDING, biased=False, n_view=None):
        '''softmax over views of the attention activation (reference network.py:354-376): (B,D,H,W,C,N) scores.'''
        import torch.nn.functional as F
        if input.dim() != 6 or biased or padding != 'SAME':
            raise NotImplementedError('attention_activation_layer: (B,D,H,W,C,N), unbiased, SAME')
        c_in, nv = input.shape[-2], input.shape[-1]
        filters = c_in if filters is None else filters
        dev = input.device
        scope = '%s/%s' % (name, name)         # the reference nests variable_scope(name) twice (:366, :314)
        wu = self.store.get('%s/weight_unique' % scope, (kernel_size,) * 3 + (c_in, filters), dev)
        act = (lambda t: torch.relu(t)) if relu else (lambda t: t)
        conv = lambda x, w: act(F.conv3d(x.permute(0, 4, 1, 2, 3), w.permute(4, 3, 0, 1, 2), padding=kernel_size // 2)  # noqa: E731
                                ).permute(0, 2, 3, 4, 1)
        xs = [input[..., n] for n in range(nv)]
        if second_weight:
            ws = self.store.get('%s/weight_shared' % scope, (kernel_size,) * 3 + (c_in, filters), dev)
            shared = [conv(x, ws) for x in xs]
            ssum = sum(shared[1:], shared[0])
            outs = [(conv(x, wu) - s) + ssum for x, s in zip(xs, shared)]
        else:
            outs = [conv(x, wu) for x in xs]
        return torch.softmax(torch.stack(outs, -1), dim=-1)

    @layer
    def attention_activation_2d(self, input, kernel_size, name, filters=None, second_weight=False, relu=True,
                                padding=DEFAULT_PADDING, biased=False):
        '''(B,H,W,C,N) -> shared-weight 2-D convolution per view -> (B,H,W,C,N) (reference network.py:411-476).'''
        import torch.nn.functional as F
        if input.dim() != 5 or biased or padding != 'SAME':
            raise NotImplementedError('attention_activation_2d: (B,H,W,C,N), unbiased, SAME')
        c_in, nv = input.shape[-2], input.shape[-1]
        filters = c_in if filters is None else filters
        dev = input.device
        wu = self.store.get('%s/weight_unique' % name, (kernel_size, kernel_size, c_in, filters), dev)
        act = (lambda t: torch.relu(t)) if relu else (lambda t: t)
        conv = lambda x, w: act(F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), padding=kernel_size // 2)  # noqa: E731
                                ).permute(0, 2, 3, 1)
        xs = [input[..., n] for n in range(nv)]
        if second_weight:
            ws = self.store.get('%s/weight_shared' % name, (kernel_size, kernel_size, c_in, filters), dev)
            shared = [conv(x, ws) for x in xs]
            ssum = sum(shared[1:], shared[0])
            outs = [(conv(x, wu) - s) + ssum for x, s in zip(xs, shared)]
        else:
            outs = [conv(x, wu) for x in xs]
        return torch.stack(outs, -1)

    @layer
    def squeeze(self, input, axis=None, name=None):
        return input.squeeze(axis) if axis is not None else input.squeeze()

    @layer
    def expand_dims(self, input, axis, name=None):
        return input.unsqueeze(axis)


def ops_b1(x, what):
    """One-sample tensor (1, ...) -> (...), contiguous."""
    if isinstance(x, (ops.SplitVolume,) + ops.LAZY):
        x = x.materialize()
    if x.shape[0] != 1:
        raise ValueError('%s: batch size must be 1 (FLAGS.batch_size), got %d' % (what, x.shape[0]))
    x = x[0]
    return x if x.is_contiguous() else x.contiguous()
