"""Variable store: weights by TensorFlow variable name.

Stands in for the TF-1.5 global variable collection that the reference restores
with ``tf.train.Saver(tf.global_variables()).restore`` (example.py:122-124).
Names follow the reference's scope rules so that a
checkpoint converted to ``{name: array}`` loads as is; layouts stay TF's
([k.., Cin, Cout]; transposed conv [k.., Cout, Cin]).

No trained checkpoint ships with the reference (.MISSING_LARGE_BLOBS), so
``init_synthetic`` fills every variable deterministically *per name*
(Xavier-normal kernels like network.py:42, N(0, 0.1) biases / BN betas so they
are exercised); creation order never changes a value.
"""
import zlib

import numpy as np

# 8->1 output convolutions get a gain so that soft-argmin sees peaked
# distributions (random-init costs are otherwise nearly flat along D)
_OUTPUT_GAIN = {
    'conv_b2_6_2/kernel': 8.0,
    'attention_prob_vol/kernel': 8.0,
    'attention_prob_vol_refine/kernel': 8.0,
    'global_refined_cost_vol/kernel': 8.0,
}


def _bottleneck_specs(scope, cin, depth, k=3):
    s = [('%s/preact/beta' % scope, (cin,))]
    if cin != depth:
        s += [('%s/shortcut/weights' % scope, (1, 1, cin, depth)), ('%s/shortcut/biases' % scope, (depth,))]
    s += [('%s/conv1/weights' % scope, (1, 1, cin, depth)), ('%s/conv1/biases' % scope, (depth,)),
          ('%s/conv2/weights' % scope, (k, k, depth, depth)), ('%s/conv2/biases' % scope, (depth,)),
          ('%s/conv3/weights' % scope, (1, 1, depth, depth)), ('%s/conv3/biases' % scope, (depth,))]
    return s


def _res_block_specs(name, cin, depth, num_block):
    scopes = [name] if num_block == 1 else \
        [name + '_0'] + [(name + '_%d' % i) if i != num_block - 1 else name for i in range(1, num_block)]
    s = []
    for i, sc in enumerate(scopes):
        s += _bottleneck_specs(sc, cin if i == 0 else depth, depth)
    return s


def variable_specs():
    """[(tf_variable_name, shape)] for every trainable variable on the hot path.

    reference graphs: cnn_wrapper/atvsnet.py (all classes used by example.py).
    """
    s = []
    # ResNetDS2SPP (atvsnet.py:254-292)
    s += [('conv0_0/conv2d/kernel', (3, 3, 3, 32)), ('conv0_1/conv2d/kernel', (3, 3, 32, 32)),
          ('conv0_2/conv2d/kernel', (3, 3, 32, 32))]
    s += _res_block_specs('conv0_x', 32, 32, 3)
    s += _res_block_specs('conv1_x', 32, 64, 8)
    s += _res_block_specs('conv2_x', 64, 128, 3)
    s += _res_block_specs('conv3_x', 128, 128, 3)
    s += [('branch_%d_conv/conv2d/kernel' % i, (3, 3, 128, 32)) for i in range(4)]
    s += [('fusion0/conv2d/kernel', (3, 3, 320, 128)), ('fusion1/kernel', (1, 1, 128, 32))]
    # ResNetDS2SPP_shallow_f16 (atvsnet.py:245-251)
    s += _res_block_specs('global_refine_conv0_x', 3, 16, 3)
    s += [('global_refine_shallow_feature/kernel', (1, 1, 16, 16))]
    # StackedUNet_prob (atvsnet.py:100-192)
    for b in range(3):
        p = 'conv_b%d_' % b
        cin = 64 if b == 0 else 8
        s += [(p + '1_0/conv3d/kernel', (3, 3, 3, cin, 16)), (p + '2_0/conv3d/kernel', (3, 3, 3, 16, 32)),
              (p + '3_0/conv3d/kernel', (3, 3, 3, 32, 64)), (p + '0_1/conv3d/kernel', (3, 3, 3, cin, 8)),
              (p + '1_1/conv3d/kernel', (3, 3, 3, 16, 16)), (p + '2_1/conv3d/kernel', (3, 3, 3, 32, 32)),
              (p + '3_1/conv3d/kernel', (3, 3, 3, 64, 64)),
              (p + '4_0/conv3d_transpose/kernel', (3, 3, 3, 32, 64)),
              (p + '5_0/conv3d_transpose/kernel', (3, 3, 3, 16, 32)),
              (p + '6_0/conv3d_transpose/kernel', (3, 3, 3, 8, 16))]
    s += [('conv_b2_6_2/kernel', (3, 3, 3, 8, 1))]
    # CostVolRefineNet (atvsnet.py:295-336)
    g = 'global_refine_'
    s += [(g + 'photo_3dconv/conv3d/kernel', (3, 3, 3, 48, 8)), (g + 'geo_3dconv/conv3d/kernel', (3, 3, 3, 19, 8)),
          (g + 'prob_3dconv/conv3d/kernel', (3, 3, 3, 1, 8)), (g + 'vishull_3dconv/conv3d/kernel', (3, 3, 3, 1, 8)),
          (g + '3dconv1_0/conv3d/kernel', (3, 3, 3, 32, 16)), (g + '3dconv2_0/conv3d/kernel', (3, 3, 3, 16, 32)),
          (g + '3dconv3_0/conv3d/kernel', (3, 3, 3, 32, 64)), (g + '3dconv0_1/conv3d/kernel', (3, 3, 3, 32, 8)),
          (g + '3dconv1_1/conv3d/kernel', (3, 3, 3, 16, 16)), (g + '3dconv2_1/conv3d/kernel', (3, 3, 3, 32, 32)),
          (g + '3dconv3_1/conv3d/kernel', (3, 3, 3, 64, 64)),
          (g + '3dconv4_0/conv3d_transpose/kernel', (3, 3, 3, 32, 64)),
          (g + '3dconv5_0/conv3d_transpose/kernel', (3, 3, 3, 16, 32)),
          (g + '3dconv6_0/conv3d_transpose/kernel', (3, 3, 3, 8, 16)),
          ('global_refined_cost_vol/kernel', (3, 3, 3, 8, 1))]
    # AANet modules + output convs (network.py:314-320; atvsnet.py:196-242)
    for a, o in (('attention_aggregate', 'attention_prob_vol'),
                 ('attention_aggregate_refine', 'attention_prob_vol_refine')):
        s += [(a + '/attention_activation/weight_unique', (3, 3, 3, 8, 8)),
              (a + '/attention_activation/weight_shared', (3, 3, 3, 8, 8)),
              (o + '/kernel', (3, 3, 3, 8, 1))]
    return s


def synthetic_value(name, shape, seed=1234):
    """Deterministic per-name value (see module docstring)."""
    rng = np.random.default_rng([int(seed), zlib.crc32(name.encode('utf-8'))])
    shape = tuple(int(v) for v in shape)
    if len(shape) == 1:
        return rng.normal(0.0, 0.1, size=shape).astype(np.float32)
    receptive = int(np.prod(shape[:-2]))
    fan_in, fan_out = shape[-2] * receptive, shape[-1] * receptive
    std = np.sqrt(2.0 / float(fan_in + fan_out))
    return (rng.normal(0.0, std, size=shape) * _OUTPUT_GAIN.get(name, 1.0)).astype(np.float32)


class VariableStore(object):
    """name -> float32 host array (authoritative) plus per-device tensor copies."""

    def __init__(self):
        self.seed = 1234
        self.host = {}
        self._dev = {}
        self.strict = False      # True: unknown names raise instead of being synthesised

    def _invalidate(self):
        """Arranged copies of the weights (ops' pack caches) are keyed by name: drop them when a value changes."""
        from . import ops
        ops.invalidate_weights()

    def clear(self):
        self.host.clear()
        self._dev.clear()
        self._invalidate()

    def init_synthetic(self, seed=1234):
        self.clear()
        self.seed = seed
        for name, shape in variable_specs():
            self.host[name] = synthetic_value(name, shape, seed)
        return self

    def set(self, name, value):
        self.host[name] = np.ascontiguousarray(value, dtype=np.float32)
        for k in [k for k in self._dev if k[0] == name]:
            del self._dev[k]
        self._invalidate()

    def load_npz(self, path):
        with np.load(path) as f:
            for k in f.files:
                self.set(k, f[k])
        self.strict = True

    def load_checkpoint(self, prefix):
        """A TensorFlow tensor-bundle checkpoint `<prefix>.index` / `<prefix>.data-*` (what the reference's
        tf.train.Saver restores, example.py:122-124): every float tensor becomes a variable under its TF name
        (optimizer slots and counters in the file are simply never asked for)."""
        from .tools import tf_checkpoint
        for k, v in tf_checkpoint.read_checkpoint(prefix).items():
            if v.dtype == np.float32:
                self.set(k, v)
        self.strict = True

    def device_snapshot(self):
        """References to every device copy handed out so far (owners of captured HIP graphs keep them alive)."""
        return list(self._dev.values())

    def save_npz(self, path):
        np.savez(path, **self.host)

    def get_host(self, name, shape):
        if name not in self.host:
            if self.strict:
                raise KeyError('variable %s not found in the loaded weights' % name)
            self.host[name] = synthetic_value(name, shape, self.seed)
        v = self.host[name]
        if tuple(v.shape) != tuple(int(s) for s in shape):
            raise ValueError('variable %s has shape %s, layer wants %s' % (name, v.shape, tuple(shape)))
        return v

    def get(self, name, shape, device):
        """Device tensor for ``name`` (created on first use, like tf.get_variable + AUTO_REUSE)."""
        import torch
        device = torch.device(device)
        key = (name, str(device))
        t = self._dev.get(key)
        if t is None:
            v = self.get_host(name, shape)
            if device.type == 'meta':
                t = torch.empty(v.shape, dtype=torch.float32, device='meta')
            else:
                t = torch.from_numpy(v).to(device)
            self._dev[key] = t
        return t


_DEFAULT = VariableStore()


def default_store():
    return _DEFAULT
