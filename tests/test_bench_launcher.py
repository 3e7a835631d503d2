"""bench.py --gpus N must start N ranks itself (the driver's command line is `python bench.py --gpus N ...`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, **extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr.decode()


def test_launcher_starts_n_ranks_dry():
    rc, line, err = _run('--gpus', '2', '--dry')
    assert rc == 0, err
    assert line['n_gpus'] == 2 and line['config']['world_size'] == 2 and line['dry'] is True
    assert line['config']['groups'] == [[0, 1]]
    # every direction of every source of the default workload (4 sources) is planned exactly once
    tasks = [tuple(t) for rank_tasks in line['config']['plan'] for t in rank_tasks]
    assert sorted(tasks) == sorted((k, v) for k in ('fwd', 'rev') for v in (1, 2, 3, 4))


def test_launcher_groups_beyond_one_rank_per_source():
    rc, line, err = _run('--gpus', '4', '--dry', '--views', '3')
    assert rc == 0, err
    assert line['n_gpus'] == 4 and line['config']['groups'] == [[0, 1], [2, 3]]


def test_launcher_refuses_a_smaller_world():
    """Fewer visible devices than --gpus: non-zero exit and a message, never a silent world of 1."""
    rc, line, err = _run('--gpus', '64', HIP_VISIBLE_DEVICES='0,1')
    assert rc != 0 and line is None
    assert 'device' in err


def test_visible_gpus_counts_without_the_gpu_runtime(tmp_path):
    """The launcher parent counts devices from the KFD topology + *_VISIBLE_DEVICES and never imports torch.cuda /
    calls HIP (a parent that opened KFD would fork+exec its ranks from a GPU-initialised process)."""
    sys.path.insert(0, ROOT)
    import bench
    nodes = tmp_path / 'nodes'
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):          # two CPU nodes, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n' % (64 if simd == 0 else 0, simd))
    assert bench.visible_gpus({}, str(nodes)) == 3
    assert bench.visible_gpus({'HIP_VISIBLE_DEVICES': '0,2'}, str(nodes)) == 2
    assert bench.visible_gpus({'ROCR_VISIBLE_DEVICES': '1', 'HIP_VISIBLE_DEVICES': '0,1,2'}, str(nodes)) == 1
    assert bench.visible_gpus({'HIP_VISIBLE_DEVICES': ''}, str(nodes)) == 0
    assert bench.visible_gpus({}, str(tmp_path / 'absent')) is None          # unknown: the ranks decide
    assert bench.visible_gpus({'CUDA_VISIBLE_DEVICES': '0,1,2,3'}, str(tmp_path / 'absent')) == 4
    src = open(os.path.join(ROOT, 'bench.py')).read()
    body = src[src.index('def launch('):src.index('# ---', src.index('def launch('))]
    assert 'torch' not in body


def test_secondary_entries_name_their_mode_and_failures():
    """What `--gpus N` adds to the primary line: every sub-object says which mode it is, carries ok / error, and the cfg4
    entry computes speed-up and fraction-of-linear from the two runs of the same invocation."""
    sys.path.insert(0, ROOT)
    import bench
    one = {'ms_per_step': 120.0, 'value': 8.33, 'parity': {'ok': True}}
    shd = {'ms_per_step': 24.0, 'value': 41.7, 'unit': 'depth-maps/sec', 'source_views_per_sec': 333.3, 'exchange': {'comm_total_ms': 3.1},
           'parity': {'ok': True, 'rel_l1': 6e-5}, 'config': {'parallelism': '1 group(s) of 8 rank(s)', 'groups': [list(range(8))]}}
    e = bench.cfg4_entry(one, None, shd, None, 8)
    assert e['ok'] and e['speedup_vs_single_gpu'] == 5.0 and e['fraction_of_linear'] == 0.625 and e['scaling'] == 'strong'
    assert e['source_views_per_sec'] == 333.3 and e['parity']['rel_l1'] == 6e-5 and 'views' in e['mode']
    bad = bench.cfg4_entry(one, None, None, 'ranks returned [1, 0]', 8)
    assert bad['ok'] is False and 'ranks returned' in bad['error'] and bad['single_gpu']['ms_per_step'] == 120.0
    v = bench.view_sharded_entry(None, 'timeout')
    assert v == {'ok': False, 'error': 'timeout'}
    l2 = dict(shd, scaling='strong')
    assert bench.view_sharded_entry(l2, None)['ok'] is True


def _fake_launch(monkeypatch, capsys, secondary_ok):
    """bench.launch() with the ranks replaced by canned lines: the primary `maps` run succeeds; the view-sharded
    secondaries succeed or fail."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []

    def fake_run_ranks(n, argv, timeout_s):
        calls.append((n, list(argv)))
        if '--parallel' in argv and argv[argv.index('--parallel') + 1] == 'maps':
            return json.dumps({'metric': 'm', 'value': 280.0, 'n_gpus': n, 'config': {}}) + '\n', [0] * n
        if not secondary_ok:
            return 'RuntimeError: NCCL error\n', [1] + [0] * (n - 1)
        cfg4 = '--workload' in argv and argv[argv.index('--workload') + 1] == 'cfg4'
        ms = (96.0 if n == 1 else 20.0) if cfg4 else 14.0
        return json.dumps({'value': 1e3 / ms, 'unit': 'depth-maps/sec', 'ms_per_step': ms, 'scaling': 'strong',
                           'source_views_per_sec': 8e3 / ms, 'exchange': {'graph_ms': ms - 3.0, 'comm_ms': 3.0},
                           'parity': {'ok': True}, 'config': {'parallelism': 'p', 'groups': [list(range(n))], 'rccl': '2.26.6',
                                                              'rccl_ranks': n}}) + '\n', [0] * n

    monkeypatch.setattr(bench, '_run_ranks', fake_run_ranks)
    monkeypatch.setattr(bench, 'visible_gpus', lambda *a, **k: 8)
    argv = ['--gpus', '8', '--steps', '3', '--warmup', '1']
    rc = bench.launch(bench.parse(argv), argv)
    out = capsys.readouterr()
    lines = [ln for ln in out.out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # ONE JSON line whatever happens
    return rc, json.loads(lines[0]), out.err, calls


def test_launcher_exits_3_when_the_view_sharded_path_breaks(monkeypatch, capsys):
    """The north-star partition must not be able to break behind a green run: a failing secondary keeps the primary line
    (value intact) but marks it ok: false and the launcher exits 3."""
    rc, line, err, calls = _fake_launch(monkeypatch, capsys, secondary_ok=False)
    assert rc == 3 and line['ok'] is False and line['value'] == 280.0
    assert set(line['failed']) == {'view_sharded', 'view_sharded_cfg4'}
    assert line['view_sharded']['ok'] is False and 'ranks returned' in line['view_sharded']['error']
    assert line['view_sharded_cfg4']['ok'] is False
    assert 'FAILED' in err and not any('fraction_of_linear' in k for k in line)


def test_launcher_promotes_the_view_sharded_numbers(monkeypatch, capsys):
    """A SCALE record reads the top level of the line: the view-sharded configs[3] rate, its fraction of linear and the
    exchange's share stand next to `value` (replica mode)."""
    rc, line, err, calls = _fake_launch(monkeypatch, capsys, secondary_ok=True)
    assert rc == 0 and 'ok' not in line or line.get('ok', True)
    assert line['value'] == 280.0
    assert line['view_sharded_cfg4_value'] == 50.0 and line['view_sharded_cfg4_ms_per_step'] == 20.0
    # promoted keys are prefixed: they belong to another mode / workload than `value`
    assert line['view_sharded_cfg4_speedup_vs_single_gpu'] == 4.8 and line['view_sharded_cfg4_fraction_of_linear'] == 0.6
    assert line['view_sharded_cfg4_exchange'] == {'graph_ms': 17.0, 'comm_ms': 3.0}
    assert line['view_sharded_cfg4_rccl_ranks'] == 8
    assert not {'fraction_of_linear', 'speedup_vs_single_gpu', 'exchange'} & set(line)
    assert line['view_sharded_cfg4_source_views_per_sec'] == 400.0
    assert abs(line['view_sharded_value'] - 1e3 / 14.0) < 1e-9
    # the three secondary runs: views at N, cfg4 on one rank, cfg4 view-sharded at N
    assert [c[0] for c in calls] == [8, 8, 1, 8]
