"""TensorFlow tensor-bundle reader (a-tvsnet_amd/tools/tf_checkpoint.py) against bundles assembled here byte by byte
from the format description (tensor_bundle.proto, table_format: prefix-compressed blocks, restart arrays, block
handles, footer magic) -- there is no real checkpoint in the reference repository to pin it with."""
import os
import struct

import numpy as np
import pytest

import atvsnet_amd  # noqa: F401
from atvsnet_amd.tools import tf_checkpoint as C


def vi(n):
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def field(num, wt, payload):
    return vi((num << 3) | wt) + (vi(len(payload)) + payload if wt == 2 else payload)


def entry_proto(dtype, shape, offset, size, shard=0):
    dims = b''.join(field(2, 2, field(1, 0, vi(d))) for d in shape)
    msg = field(1, 0, vi(dtype)) + field(2, 2, dims)
    if shard:
        msg += field(3, 0, vi(shard))
    if offset:
        msg += field(4, 0, vi(offset))
    return msg + field(5, 0, vi(size)) + field(6, 5, struct.pack('<I', 0x12345678))


def block(pairs, restart_every=2):
    """Prefix-compressed block with a restart point every `restart_every` entries + the 5-byte trailer."""
    body, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_every == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
                shared += 1
        body += vi(shared) + vi(len(k) - shared) + vi(len(v)) + k[shared:] + v
        prev = k
    body += b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
    return bytes(body), bytes(body) + b'\x00' + b'\xde\xad\xbe\xef'       # (payload, payload + trailer)


def write_bundle(prefix, tensors, n_blocks=2, num_shards=1):
    data, entries = bytearray(), []
    for name in sorted(tensors):
        arr = tensors[name]
        code = {np.dtype('float32'): 1, np.dtype('int32'): 3, np.dtype('int64'): 9}[arr.dtype]
        raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
        entries.append((name.encode(), entry_proto(code, arr.shape, len(data), len(raw))))
        data += raw
    header = field(1, 0, vi(num_shards)) + field(2, 0, vi(0)) + field(3, 2, field(1, 0, vi(1)))
    pairs = [(b'', header)] + entries
    per = (len(pairs) + n_blocks - 1) // n_blocks
    out, index_pairs = bytearray(), []
    for i in range(0, len(pairs), per):
        chunk = pairs[i:i + per]
        payload, full = block(chunk)
        index_pairs.append((chunk[-1][0] + b'~', vi(len(out)) + vi(len(payload))))
        out += full
    meta_payload, meta_full = block([])
    meta_handle = vi(len(out)) + vi(len(meta_payload))
    out += meta_full
    idx_payload, idx_full = block(index_pairs, restart_every=1)
    idx_handle = vi(len(out)) + vi(len(idx_payload))
    out += idx_full
    footer = meta_handle + idx_handle
    out += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    with open(prefix + '.index', 'wb') as f:
        f.write(out)
    with open('%s.data-00000-of-%05d' % (prefix, num_shards), 'wb') as f:
        f.write(data)


def test_read_checkpoint_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {
        'conv_b0_0_1/conv3d/kernel': rng.standard_normal((3, 3, 3, 64, 8)).astype(np.float32),
        'conv_b0_0_1/conv3d/kernel/Adam': rng.standard_normal((3, 3, 3, 64, 8)).astype(np.float32),
        'conv0_x_0/conv1/weights': rng.standard_normal((3, 3, 3, 32)).astype(np.float32),
        'conv0_x_0/conv1/biases': rng.standard_normal((32,)).astype(np.float32),
        'global_step': np.array(123456, dtype=np.int64),
        'beta1_power': np.array(0.9, dtype=np.float32),
    }
    prefix = str(tmp_path / 'model.ckpt')
    for n_blocks in (1, 3):
        write_bundle(prefix, tensors, n_blocks=n_blocks)
        got = C.read_checkpoint(prefix)
        assert sorted(got) == sorted(tensors)
        for k in tensors:
            assert got[k].dtype == tensors[k].dtype and got[k].shape == tensors[k].shape
            assert np.array_equal(got[k], tensors[k])
        assert C.list_variables(prefix)[0] == ('beta1_power', ())
        only = C.read_checkpoint(prefix, names={'conv0_x_0/conv1/biases'})
        assert list(only) == ['conv0_x_0/conv1/biases']


def test_store_loads_checkpoint_and_is_strict(tmp_path):
    from atvsnet_amd import variables
    prefix = str(tmp_path / 'm.ckpt')
    k = np.arange(3 * 3 * 3 * 8, dtype=np.float32).reshape(3, 3, 3, 8, 1)
    write_bundle(prefix, {'attention_prob_vol/kernel': k, 'global_step': np.array(7, dtype=np.int64)})
    store = variables.VariableStore()
    store.load_checkpoint(prefix)
    assert np.array_equal(store.get_host('attention_prob_vol/kernel', (3, 3, 3, 8, 1)), k)
    with pytest.raises(KeyError):
        store.get_host('conv_b0_0_1/conv3d/kernel', (3, 3, 3, 64, 8))      # not in the file: no silent random weights
    with pytest.raises(ValueError):
        store.get_host('attention_prob_vol/kernel', (3, 3, 3, 8, 2))


def test_bad_files(tmp_path):
    p = str(tmp_path / 'x')
    open(p + '.index', 'wb').write(b'\x00' * 64)
    with pytest.raises(ValueError):
        C.read_index(p)
    write_bundle(p, {'a': np.zeros((2,), np.float32)})
    os.remove(p + '.data-00000-of-00001')
    with pytest.raises(IOError):
        C.read_checkpoint(p)
    # a compressed block is refused, not mis-read
    raw = bytearray(open(p + '.index', 'rb').read())
    first_block_len = None
    _, entries = C.read_index(p)
    # flip the compression byte of the first data block (it follows the block payload: find it via the index block)
    foot = len(raw) - 48
    _, pos = C._varint(raw, foot); _, pos = C._varint(raw, pos)
    ioff, pos = C._varint(raw, pos); isize, _ = C._varint(raw, pos)
    handle = C._block(bytes(raw), ioff, isize)[0][1]
    boff, q = C._varint(handle, 0); bsize, _ = C._varint(handle, q)
    raw[boff + bsize] = 1
    open(p + '.index', 'wb').write(bytes(raw))
    with pytest.raises(NotImplementedError):
        C.read_index(p)
