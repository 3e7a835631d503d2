"""Minimal OpenEXR reader / writer: single-part scan-line images, pixel types UINT / HALF / FLOAT, compression NONE,
RLE, ZIPS and ZIP (the file-format document "OpenEXR File Layout" + "Technical Introduction to OpenEXR").

Stands in for ``imageio.imread(<depth>.exr)`` at /root/reference/atvsnet/eval_pointcloud.py:178 (the ground-truth depth
range of a scene): no OpenEXR / imageio / OpenCV exists in this image.  Tiled, multi-part, deep and PIZ / PXR24 / B44 /
DWA files raise ``ExrError`` -- the caller must fail loudly rather than silently sweep another range.
"""
import struct
import zlib

import numpy as np

MAGIC = 20000630
_PIX = {0: np.dtype('<u4'), 1: np.dtype('<f2'), 2: np.dtype('<f4')}
_LINES = {0: 1, 1: 1, 2: 1, 3: 16}
_CNAME = {0: 'NONE', 1: 'RLE', 2: 'ZIPS', 3: 'ZIP', 4: 'PIZ', 5: 'PXR24', 6: 'B44', 7: 'B44A', 8: 'DWAA', 9: 'DWAB'}


class ExrError(ValueError):
    pass


def _cstr(buf, pos):
    end = buf.index(b'\0', pos)
    return buf[pos:end].decode('latin-1'), end + 1


def _unpredict(data):
    """Undo the byte delta predictor and the even / odd byte split of the RLE / ZIP codecs."""
    a = np.frombuffer(data, np.uint8).astype(np.int64)
    if len(a) > 1:
        a[1:] -= 128
    a = (np.cumsum(a) & 255).astype(np.uint8)
    half = (len(a) + 1) // 2
    out = np.empty(len(a), np.uint8)
    out[0::2] = a[:half]
    out[1::2] = a[half:]
    return out.tobytes()


def _predict(raw):
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    return (d & 255).astype(np.uint8).tobytes()


def _unrle(data, size):
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        c = struct.unpack_from('b', data, i)[0]
        i += 1
        if c < 0:
            out += data[i:i - c]
            i += -c
        else:
            out += data[i:i + 1] * (c + 1)
            i += 1
    if len(out) != size:
        raise ExrError('RLE block decodes to %d bytes, expected %d' % (len(out), size))
    return bytes(out)


def read_header(buf):
    """-> (attributes {name: (type, raw bytes)}, position after the header)."""
    if len(buf) < 8 or struct.unpack_from('<i', buf, 0)[0] != MAGIC:
        raise ExrError('not an OpenEXR file')
    version = struct.unpack_from('<i', buf, 4)[0]
    if (version & 0xff) != 2:
        raise ExrError('OpenEXR version %d' % (version & 0xff))
    if version & (0x200 | 0x800 | 0x1000):
        raise ExrError('tiled / deep / multi-part OpenEXR files are not supported')
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        size = struct.unpack_from('<i', buf, pos)[0]
        pos += 4
        attrs[name] = (typ, buf[pos:pos + size])
        pos += size
    return attrs, pos + 1


def read_exr(path):
    """-> {channel name: (H, W) array} (float32 for HALF / FLOAT channels, uint32 for UINT)."""
    with open(path, 'rb') as f:
        buf = f.read()
    attrs, pos = read_header(buf)
    for need in ('channels', 'compression', 'dataWindow'):
        if need not in attrs:
            raise ExrError('%s: header has no %s attribute' % (path, need))
    comp = attrs['compression'][1][0]
    if comp not in _LINES:
        raise ExrError('%s: compression %s is not supported (NONE, RLE, ZIPS, ZIP are)' % (path, _CNAME.get(comp, comp)))
    x0, y0, x1, y1 = struct.unpack('<4i', attrs['dataWindow'][1][:16])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    chans, cb, p = [], attrs['channels'][1], 0
    while cb[p] != 0:
        name, p = _cstr(cb, p)
        ptype, _lin, xs, ys = struct.unpack_from('<iB3xii', cb, p)
        p += 16
        if ptype not in _PIX or xs != 1 or ys != 1:
            raise ExrError('%s: channel %s has pixel type %d / sampling %dx%d' % (path, name, ptype, xs, ys))
        chans.append((name, _PIX[ptype]))
    line_bytes = sum(dt.itemsize for _, dt in chans) * W
    lines = _LINES[comp]
    nblocks = -(-H // lines)
    offsets = struct.unpack_from('<%dQ' % nblocks, buf, pos)
    out = {name: np.zeros((H, W), np.uint32 if dt.kind == 'u' else np.float32) for name, dt in chans}
    for off in offsets:
        y, size = struct.unpack_from('<ii', buf, off)
        data = buf[off + 8:off + 8 + size]
        n = min(lines, y0 + H - y)
        want = n * line_bytes
        if comp != 0 and size < want:
            data = _unpredict(zlib.decompress(data) if comp in (2, 3) else _unrle(data, want))
        if len(data) != want:
            raise ExrError('%s: block at line %d has %d bytes, expected %d' % (path, y, len(data), want))
        p = 0
        for r in range(n):
            for name, dt in chans:
                out[name][y - y0 + r] = np.frombuffer(data, dt, W, p)
                p += dt.itemsize * W
    return out


def imread_first_channel(path):
    """What ``imageio.imread(path)[:, :, 0]`` yields for the depth EXRs of the data sets: the R channel of an RGB(A)
    file, else Y, else the first channel in the file."""
    ch = read_exr(path)
    for name in ('R', 'Y'):
        if name in ch:
            return ch[name].astype(np.float32)
    return ch[sorted(ch)[0]].astype(np.float32)


def write_exr(path, channels, compression='ZIP', half=False):
    """channels {name: (H,W) array} -> scan-line OpenEXR (FLOAT, or HALF with half=True); compression NONE | ZIPS | ZIP."""
    comp = {'NONE': 0, 'ZIPS': 2, 'ZIP': 3}[compression]
    names = sorted(channels)
    H, W = np.asarray(channels[names[0]]).shape
    dt = np.dtype('<f2') if half else np.dtype('<f4')
    chl = b''.join(n.encode('latin-1') + b'\0' + struct.pack('<iB3xii', 1 if half else 2, 0, 1, 1) for n in names) + b'\0'
    box = struct.pack('<4i', 0, 0, W - 1, H - 1)

    def attr(name, typ, data):
        return name.encode() + b'\0' + typ.encode() + b'\0' + struct.pack('<i', len(data)) + data
    head = struct.pack('<ii', MAGIC, 2)
    head += attr('channels', 'chlist', chl) + attr('compression', 'compression', bytes([comp]))
    head += attr('dataWindow', 'box2i', box) + attr('displayWindow', 'box2i', box)
    head += attr('lineOrder', 'lineOrder', b'\0') + attr('pixelAspectRatio', 'float', struct.pack('<f', 1.0))
    head += attr('screenWindowCenter', 'v2f', struct.pack('<ff', 0.0, 0.0))
    head += attr('screenWindowWidth', 'float', struct.pack('<f', 1.0)) + b'\0'
    lines = _LINES[comp]
    blocks = []
    for y in range(0, H, lines):
        raw = b''.join(np.ascontiguousarray(np.asarray(channels[n])[r], dt).tobytes()
                       for r in range(y, min(y + lines, H)) for n in names)
        data = raw
        if comp:
            z = zlib.compress(_predict(raw))
            if len(z) < len(raw):
                data = z
        blocks.append(struct.pack('<ii', y, len(data)) + data)
    pos = len(head) + 8 * len(blocks)
    table = b''
    for b in blocks:
        table += struct.pack('<Q', pos)
        pos += len(b)
    with open(path, 'wb') as f:
        f.write(head + table + b''.join(blocks))
