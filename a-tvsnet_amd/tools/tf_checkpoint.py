"""Reader for TensorFlow "tensor bundle" (V2) checkpoints -- `<prefix>.index` + `<prefix>.data-?????-of-?????` --
without TensorFlow.  The reference restores its weights with tf.train.Saver().restore(sess, model.ckpt)
(/root/reference/atvsnet/example.py:122-124, eval_pointcloud.py:274-279); this gives the same
{variable name: array} mapping to VariableStore.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table -- a LevelDB-style sorted table):
  index file  = data blocks | metaindex block | index block | footer
  footer      = BlockHandle(metaindex) BlockHandle(index), zero-padded to 40 bytes, then the magic 0xdb4775248b80fb57
  BlockHandle = varint64 offset, varint64 size (size excludes the 5-byte trailer: 1 compression byte + 4 crc bytes)
  block       = entries | uint32 restart offsets[n] | uint32 n;  entry = varint32 shared, varint32 non_shared,
                varint32 value_len, key suffix, value (keys are prefix-compressed against the previous key)
  index block : key >= last key of a data block -> BlockHandle of that block
  data blocks : key "" -> BundleHeaderProto (num_shards = 1, endianness = 2, version = 3);
                key <tensor name> -> BundleEntryProto (dtype = 1, shape = 2 {dim = 2 {size = 1}}, shard_id = 3,
                offset = 4, size = 5, crc32c = 6, slices = 7)
  data shard  = the tensors' raw bytes (little endian) at [offset, offset + size)

Checksums: every table block's trailer and every BundleEntryProto carry a MASKED crc32c (Castagnoli polynomial
0x1EDC6F41 reflected = 0x82F63B78; mask(c) = rotr(c, 15) + 0xa282ead8, tensorflow/core/lib/hash/crc32c.h) -- of the
block payload + compression byte, and of the tensor's bytes.  TensorFlow's reader fails with DataLoss on a mismatch;
so does this one (`verify=False` skips the tensor checks only).  Pinned by the standard crc32c check values
(RFC 3720 B.4) in tests/test_tf_checkpoint.py.

There is no checkpoint in the reference repository to test against: the parser is checked against bundles
assembled byte by byte from this description (tests/test_tf_checkpoint.py) -- UNPINNED against real TF output.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


_CRC_MASK_DELTA = 0xa282ead8


def _crc_tables():
    t0 = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        t0.append(c)
    tabs = [t0]
    for k in range(1, 8):
        prev = tabs[-1]
        tabs.append([(prev[i] >> 8) ^ t0[prev[i] & 0xff] for i in range(256)])
    return tabs


_T = _crc_tables()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of a bytes-like object, slicing-by-8."""
    data = bytes(data)
    c = crc ^ 0xffffffff
    n8 = len(data) // 8
    t0, t1, t2, t3, t4, t5, t6, t7 = _T
    if n8:
        for lo, hi in struct.iter_unpack('<II', data[:n8 * 8]):
            lo ^= c
            c = (t7[lo & 0xff] ^ t6[(lo >> 8) & 0xff] ^ t5[(lo >> 16) & 0xff] ^ t4[lo >> 24] ^
                 t3[hi & 0xff] ^ t2[(hi >> 8) & 0xff] ^ t1[(hi >> 16) & 0xff] ^ t0[hi >> 24])
    for b in data[n8 * 8:]:
        c = (c >> 8) ^ t0[(c ^ b) & 0xff]
    return c ^ 0xffffffff


def crc_mask(c):
    return (((c >> 15) | (c << 17)) + _CRC_MASK_DELTA) & 0xffffffff


def crc_unmask(m):
    r = (m - _CRC_MASK_DELTA) & 0xffffffff
    return ((r >> 17) | (r << 15)) & 0xffffffff


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block(buf, offset, size):
    """Decoded (key, value) pairs of one table block."""
    kind = buf[offset + size]
    if kind != 0:
        raise NotImplementedError('compressed table block (type %d): tensor-bundle indexes are written uncompressed' % kind)
    stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
    if crc_unmask(stored) != crc32c(buf[offset:offset + size + 1]):
        raise ValueError('table block at offset %d: block checksum mismatch (corrupt index file)' % offset)
    data = buf[offset:offset + size]
    n_restarts = struct.unpack_from('<I', data, len(data) - 4)[0]
    end = len(data) - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _varint(data, pos)
        fresh, pos = _varint(data, pos)
        vlen, pos = _varint(data, pos)
        key = key[:shared] + bytes(data[pos:pos + fresh])
        pos += fresh
        out.append((key, bytes(data[pos:pos + vlen])))
        pos += vlen
    return out


def _fields(msg):
    """Protobuf wire format -> list of (field number, wire type, value)."""
    pos, out = 0, []
    while pos < len(msg):
        tag, pos = _varint(msg, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(msg, pos)
        elif wt == 1:
            val, pos = msg[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(msg, pos)
            val, pos = msg[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = msg[pos:pos + 4], pos + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((num, wt, val))
    return out


def _entry(value):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'sliced': False, 'crc32c': None}
    for num, wt, val in _fields(value):
        if num == 1:
            e['dtype'] = val
        elif num == 2:
            for n2, _, dim in _fields(val):
                if n2 == 2:           # TensorShapeProto.dim
                    size = [v for k, _, v in _fields(dim) if k == 1]
                    size = size[0] if size else 0
                    e['shape'].append(size - (1 << 64) if size >= (1 << 63) else size)
        elif num == 3:
            e['shard_id'] = val
        elif num == 4:
            e['offset'] = val
        elif num == 5:
            e['size'] = val
        elif num == 6 and wt == 5:
            e['crc32c'] = struct.unpack('<I', val)[0]          # masked
        elif num == 7:
            e['sliced'] = True
    return e


def read_index(prefix):
    """-> (header dict, {tensor name: entry dict}) of `<prefix>.index`."""
    with open(prefix + '.index', 'rb') as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError('%s.index is not a TensorFlow table file (bad magic)' % prefix)
    foot = len(buf) - 48
    _, pos = _varint(buf, foot)            # metaindex handle (unused)
    _, pos = _varint(buf, pos)
    ioff, pos = _varint(buf, pos)
    isize, pos = _varint(buf, pos)
    header, entries = {'num_shards': 1, 'endianness': 0}, {}
    for _, handle in _block(buf, ioff, isize):
        boff, p = _varint(handle, 0)
        bsize, _ = _varint(handle, p)
        for key, value in _block(buf, boff, bsize):
            if key == b'':
                for num, _, val in _fields(value):
                    if num == 1:
                        header['num_shards'] = val
                    elif num == 2:
                        header['endianness'] = val
            else:
                entries[key.decode('utf-8')] = _entry(value)
    return header, entries


def list_variables(prefix):
    """[(name, shape)] like tf.train.list_variables."""
    return sorted((k, tuple(e['shape'])) for k, e in read_index(prefix)[1].items())


def read_checkpoint(prefix, names=None, verify=True):
    """{variable name: ndarray} of every (or the named) numeric tensor of the bundle `<prefix>`.  verify: check each
    tensor's bytes against the masked crc32c its index entry carries (ValueError on a mismatch, as TensorFlow's
    BundleReader returns DataLoss)."""
    header, entries = read_index(prefix)
    if header['endianness'] != 0:
        raise NotImplementedError('big-endian tensor bundle')
    shards, out = {}, {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e['sliced']:
            raise NotImplementedError('%s: partitioned (sliced) variable' % name)
        if e['dtype'] not in _DTYPES:
            continue                   # strings / resources (e.g. the saver's bookkeeping) are not weights
        sid = e['shard_id']
        if sid not in shards:
            path = '%s.data-%05d-of-%05d' % (prefix, sid, header['num_shards'])
            if not os.path.exists(path):
                raise IOError('missing data shard ' + path)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode='r')
        dt = np.dtype(_DTYPES[e['dtype']]).newbyteorder('<')
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise ValueError('%s: %d bytes stored, shape %s needs %d' % (name, e['size'], e['shape'], count * dt.itemsize))
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if len(raw) != e['size']:
            raise ValueError('%s: data shard ends before offset %d + %d' % (name, e['offset'], e['size']))
        if verify and e['crc32c'] is not None and crc_unmask(e['crc32c']) != crc32c(raw):
            raise ValueError('%s: checksum does not match the stored crc32c (corrupt data shard)' % name)
        out[name] = np.frombuffer(bytes(raw), dtype=dt).reshape(e['shape']).astype(_DTYPES[e['dtype']])
    return out
