"""Host-side data formats of the ETH3D driver (a-tvsnet_amd/atvsnet/preprocess.py) and the oracle of the
probability map: known-answer cases built by hand from the reference's definitions
(/root/reference/atvsnet/preprocess.py, model.py:13-65)."""
import io
import os
import struct

import numpy as np
import pytest
import torch

import atvsnet_amd  # noqa: F401
from atvsnet_amd import FLAGS
from atvsnet_amd.atvsnet import preprocess as P
from oracle import model as OM


@pytest.fixture(autouse=True)
def _flags():
    FLAGS.reset()
    yield
    FLAGS.reset()


CAM_HEAD = 'extrinsic\n' + '\n'.join(' '.join(str(float(4 * i + j + 1)) for j in range(4)) for i in range(4)) + \
    '\n\nintrinsic\n500.5 0.0 320.25\n0.0 501.5 240.75\n0.0 0.0 1.0\n\n'


@pytest.mark.parametrize('tail,want', [
    ('425.0 2.5', (425.0, 5.0, 64.0, 425.0 + 5.0 * 64.0)),                 # 29 words: planes = FLAGS.max_d
    ('425.0 2.5 192', (425.0, 5.0, 192.0, 425.0 + 5.0 * 192.0)),          # 30 words
    ('425.0 2.5 192 933.8', (425.0, 5.0, 192.0, 933.8)),                   # 31 words
    ('', (0.0, 0.0, 0.0, 0.0)),
])
def test_load_cam_word_counts(tail, want):
    FLAGS.max_d = 64
    cam = P.load_cam(io.StringIO(CAM_HEAD + tail + '\n'), interval_scale=2.0)
    assert cam.shape == (2, 4, 4) and cam.dtype == np.float64
    assert np.array_equal(cam[0], np.arange(1, 17, dtype=np.float64).reshape(4, 4))
    assert np.array_equal(cam[1, :3, :3], [[500.5, 0.0, 320.25], [0.0, 501.5, 240.75], [0.0, 0.0, 1.0]])
    assert np.allclose(cam[1, 3], want, rtol=0, atol=1e-12)
    assert np.all(cam[1, :3, 3] == 0)


def test_write_cam_text_and_round_trip(tmp_path):
    cam = P.load_cam(io.StringIO(CAM_HEAD + '425.0 2.5 192 933.8\n'))
    path = str(tmp_path / 'c.txt')
    P.write_cam(path, cam)
    text = open(path).read()
    assert text.startswith('extrinsic\n1.0 2.0 3.0 4.0 \n5.0 6.0 7.0 8.0 \n')
    assert '\n\nintrinsic\n500.5 0.0 320.25 \n' in text
    assert text.endswith('\n\n425.0 2.5 192.0 933.8\n')
    with open(path) as f:
        assert np.array_equal(P.load_cam(f), cam)


def test_pfm_bytes_and_round_trip(tmp_path):
    img = np.arange(6, dtype=np.float32).reshape(2, 3) + 0.5
    path = str(tmp_path / 'd.pfm')
    P.write_pfm(path, img)
    raw = open(path, 'rb').read()
    head = b'Pf\n3 2\n-1.000000\n'
    assert raw.startswith(head)
    # rows are stored bottom-up, little endian
    assert raw[len(head):] == struct.pack('<6f', 3.5, 4.5, 5.5, 0.5, 1.5, 2.5)
    with open(path, 'rb') as f:
        assert np.array_equal(P.load_pfm(f), img)
    # colour, (H,W,1) and big-endian input
    col = np.random.default_rng(0).random((4, 5, 3)).astype(np.float32)
    P.write_pfm(path, col)
    with open(path, 'rb') as f:
        assert np.array_equal(P.load_pfm(f), col)
    P.write_pfm(path, img[:, :, None])
    assert open(path, 'rb').read().startswith(b'Pf\n3 2\n')
    big = b'Pf\n2 1\n1.0\n' + struct.pack('>2f', 7.0, 8.0)
    assert np.array_equal(P.load_pfm(io.BytesIO(big)), [[7.0, 8.0]])
    with pytest.raises(Exception):
        P.write_pfm(path, img.astype(np.float64))
    with pytest.raises(Exception):
        P.load_pfm(io.BytesIO(b'P6\n1 1\n255\n'))


def test_center_image_and_scale_camera():
    img = np.random.default_rng(1).integers(0, 256, (12, 10, 3)).astype(np.uint8)
    c = P.center_image(img)
    assert c.dtype == np.float32
    assert np.allclose(c.mean(axis=(0, 1)), 0, atol=1e-5) and np.allclose(c.var(axis=(0, 1)), 1, atol=1e-4)
    cam = np.arange(32, dtype=np.float64).reshape(2, 4, 4)
    s = P.scale_camera(cam, 0.25)
    want = cam.copy()
    for r, k in ((0, 0), (1, 1), (0, 2), (1, 2)):
        want[1, r, k] *= 0.25
    assert np.array_equal(s, want) and s is not cam


def test_scale_image_sampling():
    # scale 1 is the identity; 0.25 averages the two middle pixels of every group of four (uint8 rounding to nearest)
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (16, 24, 3)).astype(np.uint8)
    assert np.array_equal(P.scale_image(img, 1), img)
    q = P.scale_image(img, 0.25)
    assert q.shape == (4, 6, 3) and q.dtype == np.uint8
    mid = (img[1::4].astype(np.float64) + img[2::4]) / 2
    want = (mid[:, 1::4] + mid[:, 2::4]) / 2
    assert np.abs(q.astype(np.float64) - want).max() <= 0.75        # exact value up to fixed-point rounding
    # float images: plain bilinear, exact for a linear ramp away from the border
    ramp = (np.arange(40, dtype=np.float32)[None, :] * 2.0 + np.arange(30, dtype=np.float32)[:, None] * 3.0)
    half = P.scale_image(ramp, 0.5)
    assert half.shape == (15, 20)
    yy, xx = np.meshgrid(np.arange(15) * 2 + 0.5, np.arange(20) * 2 + 0.5, indexing='ij')
    assert np.allclose(half, xx * 2.0 + yy * 3.0, atol=1e-4)
    # upscaling clamps at the border; nearest picks floor(i / scale)
    up = P.scale_image(ramp[:2, :2], 2.0)
    assert up.shape == (4, 4) and up[0, 0] == ramp[0, 0] and up[3, 3] == ramp[1, 1]
    near = P.scale_image(np.tile(np.arange(8, dtype=np.float32)[None, :], (2, 1)), 0.5, interpolation='nearest')
    assert np.array_equal(near, [[0, 2, 4, 6]])


def test_crop_mvs_input_windows_and_principal_point():
    FLAGS.view_num, FLAGS.max_h, FLAGS.max_w = 2, 64, 96
    images = [np.zeros((70, 100, 3), np.uint8), np.zeros((60, 90, 3), np.uint8)]
    for im in images:
        im[...] = np.arange(im.shape[1], dtype=np.uint8)[None, :, None]
    cams = [np.zeros((2, 4, 4)), np.zeros((2, 4, 4))]
    for c in cams:
        c[1, 0, 2], c[1, 1, 2] = 50.0, 35.0
    out, oc = P.crop_mvs_input(images, cams, base_image_size=32)
    # view 0: 70x100 -> capped 64x96, start = ceil(6/2) = 3, ceil(4/2) = 2
    assert out[0].shape == (64, 96, 3) and out[0][0, 0, 0] == 2
    assert oc[0][1, 0, 2] == 48.0 and oc[0][1, 1, 2] == 32.0
    # view 1: 60x90 -> rounded up to 64x96 (no pixels to add: the slice keeps what exists), start = ceil(-2) = -2 / -3
    assert oc[1][1, 0, 2] == 50.0 - int(np.ceil((90 - 96) / 2)) and oc[1][1, 1, 2] == 35.0 - int(np.ceil((60 - 64) / 2))


def test_mask_depth_image():
    d = np.array([[0.5, 1.0, 2.0, 3.0, 4.0]], np.float32)
    m = P.mask_depth_image(d, 1.0, 3.0)
    assert m.shape == (1, 5, 1) and np.array_equal(m[0, :, 0], [0, 0, 2.0, 3.0, 0])


def test_pair_list(tmp_path):
    FLAGS.view_num = 3
    (tmp_path / 'pair.txt').write_text('2\n0\n3 5 0.9 7 0.8 9 0.7\n5\n1 0 0.5\n')
    got = P.gen_pipeline_mvs_list(str(tmp_path))
    d = str(tmp_path)
    img = lambda i: os.path.join(d, 'images', '%08d.jpg' % i)          # noqa: E731
    cam = lambda i: os.path.join(d, 'cams', '%08d_cam.txt' % i)        # noqa: E731
    assert got == [[img(0), cam(0), img(5), cam(5), img(7), cam(7)], [img(5), cam(5), img(0), cam(0)]]


def test_oracle_probability_map_known_answer():
    # D = 4 planes with probabilities per pixel; depth_start 1, interval 0.5
    cv = torch.tensor([[0.1, 0.1], [0.2, 0.3], [0.3, 0.4], [0.4, 0.2]]).reshape(1, 4, 1, 2)
    ds, di = torch.tensor([1.0]), torch.tensor([0.5])
    # pixel 0: d = (1.6-1)/.5 = 1.2 -> l0 1, l1 0, r0 2, r1 3 -> 0.2+0.1+0.3+0.4 = 1.0
    # pixel 1: d = (2.0-1)/.5 = 2.0 (integral) -> l0 2, l1 1, r0 2, r1 3 -> 0.4+0.3+0.4+0.2 = 1.3
    depth = torch.tensor([1.6, 2.0]).reshape(1, 1, 2, 1)
    got = OM.get_propability_map(cv, depth, ds, di)
    assert torch.allclose(got.reshape(-1), torch.tensor([1.0, 1.3]), atol=1e-6)
    # below / above the sweep: everything clips to the first / last plane
    depth = torch.tensor([0.0, 9.0]).reshape(1, 1, 2, 1)
    got = OM.get_propability_map(cv, depth, ds, di)
    # pixel 0: d=-2 -> l0 0, l1 0, r0 0, r1 1 -> .1+.1+.1+.2 ; pixel 1: d=16 -> 3,2,3,3 -> .2+.4+.2+.2
    assert torch.allclose(got.reshape(-1), torch.tensor([0.5, 1.0]), atol=1e-6)
    # the softmax probabilities sum to 1: a 4-plane volume gives prob 1 + p(double-counted) at most
    v = torch.randn(1, 4, 3, 3)
    d, p = OM.prob2depth(v, 4, ds, di, True)
    assert p.shape == (1, 3, 3, 1) and float(p.min()) > 0 and float(p.max()) <= 4.0


def _hand_made_zip_exr(path, depth):
    """An RGB float scan-line EXR with ZIP blocks assembled here byte by byte from the file-layout document (NOT with
    tools.exr.write_exr): header attributes, offset table, 16-line blocks = zlib(delta-predicted(even bytes | odd bytes))."""
    import zlib
    H, W = depth.shape
    att = lambda n, t, d: n.encode() + b'\0' + t.encode() + b'\0' + struct.pack('<i', len(d)) + d      # noqa: E731
    chl = b''.join(c + b'\0' + struct.pack('<i', 2) + bytes([0, 0, 0, 0]) + struct.pack('<ii', 1, 1) for c in (b'B', b'G', b'R')) + b'\0'
    box = struct.pack('<4i', 0, 0, W - 1, H - 1)
    head = struct.pack('<i', 20000630) + struct.pack('<i', 2)
    head += att('channels', 'chlist', chl) + att('compression', 'compression', b'\x03') + att('dataWindow', 'box2i', box)
    head += att('displayWindow', 'box2i', box) + att('lineOrder', 'lineOrder', b'\x00')
    head += att('pixelAspectRatio', 'float', struct.pack('<f', 1.0)) + att('screenWindowCenter', 'v2f', struct.pack('<ff', 0, 0))
    head += att('screenWindowWidth', 'float', struct.pack('<f', 1.0)) + b'\0'
    blocks = []
    for y in range(0, H, 16):
        raw = bytearray()
        for r in range(y, min(y + 16, H)):
            for k in (3.0, 2.0, 1.0):                       # B, G, R planes of the line: R = depth
                raw += (depth[r] * np.float32(k)).astype('<f4').tobytes()
        t = bytes(raw[0::2]) + bytes(raw[1::2])
        d = bytearray(t)
        for i in range(len(t) - 1, 0, -1):
            d[i] = (t[i] - t[i - 1] + 128) & 255
        z = zlib.compress(bytes(d))
        assert len(z) < len(raw)                                  # (a block that does not shrink is stored raw)
        blocks.append(struct.pack('<ii', y, len(z)) + z)
    pos = len(head) + 8 * len(blocks)
    table = b''
    for b in blocks:
        table += struct.pack('<Q', pos)
        pos += len(b)
    with open(path, 'wb') as f:
        f.write(head + table + b''.join(blocks))


def test_exr_reader_and_ground_truth_depth_range(tmp_path):
    """tools/exr.py on a ZIP file assembled by hand, and the ground-truth range branch of the ETH3D driver
    (reference eval_pointcloud.py:170-192) that uses it."""
    from atvsnet_amd.atvsnet import eval_pointcloud as EP
    from atvsnet_amd.tools import exr
    rng = np.random.default_rng(5)
    yy, xx = np.meshgrid(np.arange(37), np.arange(21), indexing='ij')
    depth = (2.0 + 0.25 * (xx + 2 * yy) + rng.integers(0, 2, (37, 21)) * 0.125).astype(np.float32)    # compressible
    depth[3, 4] = 0.0                                           # a hole: excluded from the range
    scene = tmp_path / 'scene'
    (scene / 'images').mkdir(parents=True)
    (scene / 'depths').mkdir()
    _hand_made_zip_exr(str(scene / 'depths' / 'orig_0007.exr'), depth)
    ch = exr.read_exr(str(scene / 'depths' / 'orig_0007.exr'))
    assert sorted(ch) == ['B', 'G', 'R'] and np.array_equal(ch['R'], depth) and np.array_equal(ch['B'], depth * np.float32(3.0))
    assert np.array_equal(exr.imread_first_channel(str(scene / 'depths' / 'orig_0007.exr')), depth)
    # the writer (used by nothing but tests) agrees with the hand-made file
    exr.write_exr(str(tmp_path / 'w.exr'), {'R': depth, 'G': depth * np.float32(2.0), 'B': depth * np.float32(3.0)}, 'ZIP')
    assert np.array_equal(exr.read_exr(str(tmp_path / 'w.exr'))['G'], depth * np.float32(2.0))
    with pytest.raises(exr.ExrError):
        open(str(tmp_path / 'bad.exr'), 'wb').write(b'not an exr file at all')
        exr.read_exr(str(tmp_path / 'bad.exr'))
    # the driver: <ref>.txt names the original image, whose /images/ -> /depths/ .exr gives the sweep of every view
    ref = scene / 'images' / '00000000.jpg'
    ref.write_bytes(b'')
    (scene / 'images' / '00000000.txt').write_text('orig_0007.png\n')
    FLAGS.max_d, FLAGS.inverse_depth = 128, True
    cams = [np.zeros((2, 4, 4)) for _ in range(3)]
    EP._ground_truth_depth_range(str(ref), cams)
    valid = depth[depth > 0]
    dmin, dmax = 1.0 / float(valid.max()), 1.0 / float(valid.min())
    for cam in cams:
        assert abs(cam[1][3][0] - dmin) < 1e-7 and abs(cam[1][3][3] - dmax) < 1e-7
        assert cam[1][3][2] == 128 and abs(cam[1][3][1] - (dmax - dmin) / 128) < 1e-9
    # no note, or no EXR behind it: the camera-file range stays
    cams2 = [np.ones((2, 4, 4))]
    EP._ground_truth_depth_range(str(scene / 'images' / '00000009.jpg'), cams2)
    (scene / 'images' / '00000001.txt').write_text('missing.png\n')
    EP._ground_truth_depth_range(str(scene / 'images' / '00000001.jpg'), cams2)
    assert np.array_equal(cams2[0], np.ones((2, 4, 4)))
    # an EXR that exists but cannot be decoded must raise, not change the sweep silently
    (scene / 'depths' / 'broken.exr').write_bytes(b'\x76\x2f\x31\x01' + b'\x02\x00\x00\x00' + b'garbage')
    (scene / 'images' / '00000002.txt').write_text('broken.png\n')
    with pytest.raises(Exception):
        EP._ground_truth_depth_range(str(scene / 'images' / '00000002.jpg'), cams2)
