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


def crc32c_bitwise(data):
    """Independent bit-at-a-time CRC-32C (the writer side of these tests; the reader uses slicing-by-8 tables)."""
    c = 0xffffffff
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xffffffff


def masked(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


def entry_proto(dtype, shape, offset, size, shard=0, crc=0):
    dims = b''.join(field(2, 2, field(1, 0, vi(d))) for d in shape)
    msg = field(1, 0, vi(dtype)) + field(2, 2, dims)
    if shard:
        msg += field(3, 0, vi(shard))
    if offset:
        msg += field(4, 0, vi(offset))
    return msg + field(5, 0, vi(size)) + field(6, 5, struct.pack('<I', masked(crc)))


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
    trailer = struct.pack('<I', masked(crc32c_bitwise(bytes(body) + b'\x00')))
    return bytes(body), bytes(body) + b'\x00' + trailer                   # (payload, payload + type byte + crc)


def write_bundle(prefix, tensors, n_blocks=2, num_shards=1):
    data, entries = bytearray(), []
    for name in sorted(tensors):
        arr = tensors[name]
        code = {np.dtype('float32'): 1, np.dtype('int32'): 3, np.dtype('int64'): 9}[arr.dtype]
        raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
        entries.append((name.encode(), entry_proto(code, arr.shape, len(data), len(raw), crc=crc32c_bitwise(raw))))
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


def test_crc32c_known_answers():
    """RFC 3720 appendix B.4 check values + the classic '123456789' vector; mask / unmask as crc32c.h defines them."""
    assert C.crc32c(b'123456789') == 0xE3069283
    assert C.crc32c(b'\x00' * 32) == 0x8A9136AA
    assert C.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert C.crc32c(bytes(range(32))) == 0x46DD794E
    assert C.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert C.crc32c(b'') == 0
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 9, 63, 1000):
        buf = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert C.crc32c(buf) == crc32c_bitwise(buf)
        assert C.crc32c(buf[3:], C.crc32c(buf[:3])) == C.crc32c(buf)           # incremental form
    for c in (0, 1, 0xE3069283, 0xffffffff):
        assert C.crc_unmask(C.crc_mask(c)) == c and C.crc_mask(c) == masked(c)
    assert C.crc_mask(0xE3069283) != 0xE3069283


def test_corrupt_bytes_raise(tmp_path):
    """One flipped byte in a tensor or in an index block is an error (TensorFlow: DataLoss), never silently different
    weights."""
    p = str(tmp_path / 'c.ckpt')
    w = np.arange(64, dtype=np.float32)
    write_bundle(p, {'w': w, 'v': np.ones((3,), np.float32)})
    assert np.array_equal(C.read_checkpoint(p)['w'], w)
    shard = p + '.data-00000-of-00001'
    raw = bytearray(open(shard, 'rb').read())
    raw[17] ^= 0x40
    open(shard, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        C.read_checkpoint(p)
    assert not np.array_equal(C.read_checkpoint(p, verify=False)['w'], w)       # explicit opt-out still reads
    assert np.array_equal(C.read_checkpoint(p, names={'w'}, verify=False)['w'].shape, w.shape)
    raw[17] ^= 0x40
    open(shard, 'wb').write(bytes(raw))
    idx = bytearray(open(p + '.index', 'rb').read())
    idx[5] ^= 0x01                                     # inside the first data block
    open(p + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='checksum'):
        C.read_index(p)


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
