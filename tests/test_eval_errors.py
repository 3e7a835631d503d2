"""CPU: calc_error (product mirror and oracle restatement) against golden vectors generated from
the reference's own function (tests/golden/make_calc_error_golden.py) and the numbers in the
reference's example/*/result/error.xlsx; the xlsx writer round-trips."""
import os

import numpy as np
import pytest

from oracle import eval_errors as OE

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _impls():
    from atvsnet_amd.atvsnet import eval_errors as PE
    return [('product', PE.calc_error), ('oracle', OE.calc_error)]


@pytest.mark.parametrize('idx', [0, 1, 2])
def test_calc_error_golden(idx):
    g = np.load(os.path.join(GOLD, 'calc_error_golden.npz'))
    for name, fn in _impls():
        e, info = fn(g['pred_%d' % idx], g['gt_%d' % idx])
        assert e.dtype == np.float32 and e.shape == (14,)
        assert np.allclose(e, g['err_%d' % idx], rtol=1e-6, atol=0), name
        assert np.allclose(np.array(info[:4], np.float64), g['info_%d' % idx], rtol=1e-6), name


def test_calc_error_edge_cases():
    g = np.load(os.path.join(GOLD, 'calc_error_golden.npz'))
    for name, fn in _impls():
        e, _ = fn(g['pred_edge'], g['gt_edge'], num_depths=64, inlier_threshold=[1, 2, 4])
        assert e.shape == (13,) and np.allclose(e, g['err_edge'], rtol=1e-6), name


def test_inputs_are_not_modified():
    g = np.load(os.path.join(GOLD, 'calc_error_golden.npz'))
    p, t = g['pred_edge'].copy(), g['gt_edge'].copy()
    for _, fn in _impls():
        fn(p, t)
        assert np.array_equal(p, g['pred_edge'], equal_nan=True) and np.array_equal(t, g['gt_edge'], equal_nan=True)


@pytest.mark.parametrize('idx,views', [(0, 5), (1, 5), (2, 2)])
def test_reference_xlsx_known_answers(idx, views):
    """The full-size metrics recomputed by the reference function equal the xlsx the authors shipped;
    our xlsx reader parses it (sheet '<views>_view', layout of example.py:199-213)."""
    from atvsnet_amd.tools import xlsx
    from atvsnet_amd.atvsnet.eval_errors import acc_metrics_namelist, err_metrics_namelist
    name, cells = xlsx.read_xlsx(os.path.join(GOLD, 'example%d_error.xlsx' % idx))
    assert name == '%d_view' % views
    full = np.load(os.path.join(GOLD, 'calc_error_golden.npz'))['full_%d' % idx]
    assert cells[(0, 1)] == 'err' and cells[(11, 1)] == 'acc'
    for i, m in enumerate(err_metrics_namelist):
        assert cells[(i + 1, 0)] == m
        assert cells[(i + 1, 1)] == pytest.approx(float(full[i]), rel=1e-6)
    for i, m in enumerate(acc_metrics_namelist):
        assert cells[(i + 12, 0)] == m
        assert cells[(i + 12, 1)] == pytest.approx(float(full[10 + i]), rel=1e-6)


def test_write_error_xlsx_roundtrip(tmp_path):
    from atvsnet_amd.atvsnet import example as ex
    from atvsnet_amd.tools import xlsx
    err = np.load(os.path.join(GOLD, 'calc_error_golden.npz'))['full_2']
    path = str(tmp_path / 'error.xlsx')
    ex.write_error_xlsx(path, err, 2)
    name, cells = xlsx.read_xlsx(path)
    ref_name, ref_cells = xlsx.read_xlsx(os.path.join(GOLD, 'example2_error.xlsx'))
    assert name == ref_name and set(cells) == set(ref_cells)
    for k, v in ref_cells.items():
        if isinstance(v, str):
            assert cells[k] == v
        else:
            assert cells[k] == pytest.approx(v, rel=1e-6)
