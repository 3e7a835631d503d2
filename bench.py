#!/usr/bin/env python
"""Headline benchmark: depth-maps/sec of the A-TVSNet multi-view inference path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5]

A step = one depth map through the whole example.py pipeline (towers -> 2x stacked 3-D U-Net per source ->
AAM1 -> refinement per source -> AAM2 -> x4 upsample + soft-argmin), inputs resident in HBM, synthetic
seeded images / cameras / weights, fp32 storage and accumulation; the heavy convolutions run on
v_mfma_f32_16x16x32_f16 with every fp32 operand split into two fp16 pieces (three products, fp32-class results),
the rest on the fp32 matrix cores; `split_operands` times the
all-fp32-MFMA path next to it.
Default workload = BASELINE.json configs[2], the configuration the metric is quoted on: 5 views
(1 reference + 4 sources) of 640x512, D=192.  cfg2 = two-view 640x512x192, cfg4 = 9 views 928x480x256 (the
8-source configuration quoted for 8 GPUs), cfg5 = two-view 1600x1184x256.
On one GPU the step is ONE replay of a HIP graph captured from the pipeline (every per-view network evaluated once
over all its calls -- views, siamese directions -- stacked on the batch axis with per-call batch statistics);
--eager issues every launch from Python instead.  `value` = K depth maps strictly one after the other (1 / wall
time of one depth map).  --inflight N (default 2): a SECOND timed region issues K depth maps round-robin on N streams
with their kernels CO-RESIDENT on the GPU (example.PipelinedInference(co_resident=True), opt-in in the product) -- the
depth maps of a scene are independent (one per reference view) and a second one in flight fills the phases in which one
pipeline leaves the GPU under-filled; that throughput is reported under `pipelined`, never as `value`, and the run FAILS
(exit code 4, "ok": false) if any slot's depth map differs from the single-map output by one bit.
A THIRD region repeats it with every map confined to its own share of every XCD's compute units (hipExtStreamCreateWithCUMask,
example.PipelinedInference(co_resident='cu_split')): kernels of different maps then never share a SIMD -- the safe form of maps in
flight -- reported under `pipelined_cu_split`, same bitwise check.

--gpus N > 1: this process touches no GPU; it starts N ranks (one process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous), prints ONE JSON line and exits non-zero if a rank of the primary
measurement fails (1), if fewer than N devices are visible (2), or if a secondary view-sharded measurement fails (3: the
line is still printed, with "ok": false and the error inside the sub-object).  (Launched under torch.distributed.run -- WORLD_SIZE
already in the environment -- it is a rank itself and measures the mode --parallel names.)  Two ways to use N GPUs
(a-tvsnet_amd/parallel.py):
  --parallel maps   (primary; `value`): the unit of the metric is a depth map and the depth maps of a scene are
                    independent (one per reference view, reference eval_pointcloud.py:399-424): every rank computes
                    its own depth maps, no data-path collective, "scaling": "weak".
  --parallel views  (secondary; reported under "view_sharded" of the same line): the source views of ONE depth map
                    are sharded over groups of at most one rank per source and exchanged inside both AANet modules
                    over RCCL (all-to-all of voxel shards + all-gather) -- lower latency per depth map, fewer depth
                    maps per second than `maps` (DESIGN.md 5 has the model).  Measured by a second set of ranks after
                    the primary result is safe; a failure there is reported in the line, on stderr and as exit code 3.
With the default workload the launcher ALSO measures BASELINE configs[3] ("view_sharded_cfg4": 9 views 928x480x256, the 8
sources dealt over the N ranks -- one per GPU at N = 8 -- with `source_views_per_sec`, the exchange's share, the parity
against tests/golden/fullsize_cfg4.npz, and the same depth map on ONE rank measured in the same invocation, so that
speed-up and fraction-of-linear come from one run).

One JSON line on rank 0.  `roofline` = the dominant kernel (conv_xb.hip: the 3x3x3 convolution of the 32 warped
channels of conv_b0_0_1 at full resolution together with its stride-2 sibling conv_b0_1_0, one launch): its ALGORITHMIC
convolution FLOPs / its duration (HIP events on its launch stream, >= 10 eager launches) against the dense bf16 /
fp16 MFMA peak (2.5 PF) -- `frac` = `algorithmic_frac`; `emulation_ceiling_frac` = the same against peak / 3 (three products per
fp32-accurate product); `mfma_pipe_utilisation` = the MFMA FLOPs the launch ISSUES against the peak (how busy the pipe is, not how
much useful work it does); `fp32_equivalent` = the algorithmic FLOPs against the fp32 matrix peak, `mfma_busy` / `clock_GHz` from the
committed PMC passes; `roofline_hbm` = the plane-sweep warp
(warp_planes_shared_kernel) against the HBM peak, timed the same way; `kernels` = the top kernels of the committed
rocprofv3 kernel trace of this command; `parity` = the output of the timed path against the oracle-generated
fixture of this workload (tests/golden/fullsize_*.npz) and graph replay == eager bit for bit;
`cpu_baseline` = the CPU oracle on the host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {          # name: (views, width, height, depths)
    'cfg2': (2, 640, 512, 192),
    'cfg3': (5, 640, 512, 192),
    'cfg4': (9, 928, 480, 256),
    'cfg5': (2, 1600, 1184, 256),
}
DOMINANT = ('conv_b0_0_1/conv3d/kernel', 'var')   # the D-varying half of conv_b0_0_1 (see ops.conv_split)
WARP = ('warp', 0)                                 # atvs_warp_planes, bilinear (the cost-volume build)
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md, dense BF16 MFMA
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md, HBM3E peak (6.29 TB/s achievable by a float4 copy)
def _newest(*names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, 'profiles', n)):
            return os.path.join('profiles', n)
    return os.path.join('profiles', names[-1])


PMC_FILE = _newest('round6_pmc_xpair.json', 'round5_pmc_xpair.json', 'round4_pmc_xpair.json', 'round3_pmc_xpair.json')
PMC_KERNELS_FILE = _newest('round6_pmc_kernels.json', 'round5_pmc_kernels.json')
KERNEL_STATS_FILE = _newest('round6_bench_kernel_stats.csv', 'round5_bench_kernel_stats.csv', 'round4_bench_kernel_stats.csv',
                            'round3_bench_kernel_stats.csv')


def profile_head(path):
    """The `git rev-parse HEAD` a committed profile was measured at (tools_dev/final_run.sh writes profiles/<name>.head next to
    every summary it copies; None for the profiles of earlier rounds)."""
    try:
        with open(os.path.join(ROOT, os.path.splitext(path)[0] + '.head')) as f:
            return f.read().strip() or None
    except OSError:
        return None
EAGER_TIMING_PASSES = 10           # eager passes after the timed region: >= 10 event-timed launches of the dominant kernel


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--workload', choices=sorted(WORKLOADS), default='cfg3')
    p.add_argument('--width', type=int, default=None)
    p.add_argument('--height', type=int, default=None)
    p.add_argument('--depths', type=int, default=None)
    p.add_argument('--views', type=int, default=None)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-power', action='store_true', help='skip the ~2 s socket-power / shader-clock probe after the timed region')
    p.add_argument('--no-parity', action='store_true')
    p.add_argument('--no-fp32-path', '--no-split-bf16', dest='no_fp32_path', action='store_true',
                   help='skip the secondary measurement of the same step with every convolution on the fp32 matrix cores')
    p.add_argument('--eager', action='store_true', help='issue every launch from Python instead of replaying a HIP graph')
    p.add_argument('--inflight', type=int, default=2,
                   help='depth maps in flight per GPU (streams with one captured graph each); 1 = strictly one after the other')
    p.add_argument('--parallel', choices=['maps', 'views', 'both'], default=None,
                   help='N > 1: maps = one depth map per rank (no collective), views = source views sharded inside groups; '
                        'the launcher default measures maps, then views as a secondary result of the same line')
    p.add_argument('--split-directions', action='store_true',
                   help='with >= 2 ranks per source: one siamese direction per rank instead of further depth-map groups')
    p.add_argument('--dry', action='store_true',
                   help='no GPU: the ranks rendezvous over gloo, agree on the plan and print the line (launcher test)')
    a = p.parse_args(argv)
    v, w, h, d = WORKLOADS[a.workload]
    a.custom = any(x is not None for x in (a.width, a.height, a.depths, a.views))
    a.views = a.views or v
    a.width = a.width or w
    a.height = a.height or h
    a.depths = a.depths or d
    return a


# --------------------------------------------------------------------------------------------- launcher

def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(n, argv, timeout_s):
    """Start n ranks of this script; -> (rank 0's stdout text, return codes).  Children that outlive the timeout are
    killed (the exact processes started here)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    deadline = time.time() + timeout_s
    out0 = b''
    try:
        out0, _ = procs[0].communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        pass
    rcs = []
    for p in procs:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()          # the exact child we started
            p.wait()
        rcs.append(p.returncode)
    return (out0.decode('utf-8', 'replace') if out0 else ''), rcs


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith('{')]
    return json.loads(lines[-1]) if lines else None


def visible_gpus(env=None, kfd_nodes='/sys/class/kfd/kfd/topology/nodes'):
    """Number of GPUs the ranks will see, WITHOUT opening the GPU runtime in this process (no HIP call, no torch.cuda:
    on ROCm wheels without amdsmi torch.cuda.device_count() falls through to hipGetDeviceCount, which opens KFD).
    KFD topology nodes with simd_count > 0 are GPUs; HIP_/ROCR_/CUDA_VISIBLE_DEVICES restrict them.  None when the
    topology is not readable (then the ranks' own `device_count() > local_rank` check decides)."""
    env = os.environ if env is None else env
    total = None
    try:
        total = 0
        for node in os.listdir(kfd_nodes):
            try:
                with open(os.path.join(kfd_nodes, node, 'properties')) as f:
                    props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            except OSError:
                continue
            if int(props.get('simd_count', '0')) > 0:
                total += 1
    except OSError:
        total = None
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if var in env:
            ids = [t for t in env[var].split(',') if t.strip() != '']
            total = len(ids) if total is None else min(total, len(ids))
    return total


def launch(args, argv):
    """Start args.gpus ranks of this script as child processes; no GPU call happens in this process."""
    n = args.gpus
    if not args.dry:
        have = visible_gpus()
        if have is not None and have < n:
            sys.stderr.write('bench.py: --gpus %d but only %d device(s) visible; refusing to run a smaller world\n' % (n, have))
            return 2
    base = [a for i, a in enumerate(argv) if a != '--parallel' and (i == 0 or argv[i - 1] != '--parallel')]
    first = 'views' if args.parallel == 'views' else 'maps'
    text, rcs = _run_ranks(n, base + ([] if args.dry else ['--parallel', first]), 1500)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    line = _last_json(text)
    if bad or line is None:
        sys.stdout.write(text)
        sys.stderr.write('bench.py: rank(s) failed: %s\n' % bad)
        return 1
    if not args.dry and args.parallel in (None, 'both'):
        # secondary results of the same line, measured after the primary one is safe (a failure / hang there is reported
        # inside the line and on stderr, it does not cost the primary number):
        #   view_sharded       the same workload with its source views sharded inside groups (RCCL exchange in both AANets)
        #   view_sharded_cfg4  BASELINE configs[3]: 9 views 928x480x256, its 8 sources dealt over the N ranks (N = 8: one
        #                      per GPU), next to the same depth map on ONE rank measured in the same invocation
        def secondary(nranks, extra, timeout_s):
            try:
                t2, rc2 = _run_ranks(nranks, extra + ['--no-cpu-baseline'], timeout_s)
                l2 = _last_json(t2)
                if l2 is not None and all(rc == 0 for rc in rc2):
                    return l2, None
                return None, 'ranks returned %s' % rc2
            except Exception as e:
                return None, repr(e)
        if args.views > 2:
            l2, err = secondary(n, base + ['--parallel', 'views'], 600)
            line['view_sharded'] = view_sharded_entry(l2, err)
        if not args.custom and args.workload == 'cfg3':
            wl = [a for i, a in enumerate(base) if a != '--workload' and (i == 0 or base[i - 1] != '--workload')]
            wl = [a for i, a in enumerate(wl) if a not in ('--gpus',) and (i == 0 or wl[i - 1] != '--gpus')]
            one, err1 = secondary(1, wl + ['--gpus', '1', '--workload', 'cfg4', '--inflight', '1', '--steps', '3', '--warmup', '1'], 900)
            shd, err2 = secondary(n, wl + ['--gpus', str(n), '--workload', 'cfg4', '--parallel', 'views', '--steps', '5', '--warmup', '2'], 900)
            line['view_sharded_cfg4'] = cfg4_entry(one, err1, shd, err2, n)
        failed = []
        for key in ('view_sharded', 'view_sharded_cfg4'):
            if isinstance(line.get(key), dict) and not line[key].get('ok', False):
                failed.append(key)
                sys.stderr.write('bench.py: secondary measurement %s FAILED: %s\n' % (key, line[key].get('error')))
        promote_view_sharded(line)
        if failed:
            line['ok'] = False
            line['failed'] = failed
            print(json.dumps(line))
            sys.stdout.flush()
            return 3
    print(json.dumps(line))
    sys.stdout.flush()
    return 0


def promote_view_sharded(line):
    """The north-star partition (BASELINE configs[3]: source views sharded one per GPU, exchange inside both AANets) next
    to `value` at the top level of the N > 1 line, so that a SCALE record shows it and not only the replica mode.  Every
    promoted key carries the `view_sharded_cfg4_` / `view_sharded_` prefix: they belong to another mode and workload than `value`."""
    c4 = line.get('view_sharded_cfg4')
    if isinstance(c4, dict) and c4.get('ok'):
        line['view_sharded_cfg4_value'] = c4.get('value')
        line['view_sharded_cfg4_ms_per_step'] = c4.get('ms_per_step')
        line['view_sharded_cfg4_source_views_per_sec'] = c4.get('source_views_per_sec')
        line['view_sharded_cfg4_fraction_of_linear'] = c4.get('fraction_of_linear')
        line['view_sharded_cfg4_speedup_vs_single_gpu'] = c4.get('speedup_vs_single_gpu')
        line['view_sharded_cfg4_exchange'] = c4.get('exchange')
        line['view_sharded_cfg4_rccl_ranks'] = c4.get('rccl_ranks')
    vs = line.get('view_sharded')
    if isinstance(vs, dict) and vs.get('ok'):
        line['view_sharded_value'] = vs.get('value')
    return line


def view_sharded_entry(l2, err):
    """The view-sharded run of the primary workload as a sub-object of the primary line."""
    if l2 is None:
        return {'ok': False, 'error': err}
    out = {k: l2.get(k) for k in ('value', 'unit', 'ms_per_step', 'scaling', 'source_views_per_sec', 'exchange', 'parity')}
    out.update(ok=True, mode='views (source views of one depth map sharded inside a group, RCCL exchange in AAM1/AAM2)',
               parallelism=l2['config']['parallelism'], groups=l2['config']['groups'], rccl=l2['config'].get('rccl'),
               rccl_ranks=l2['config'].get('rccl_ranks'))
    return out


def cfg4_entry(one, err1, shd, err2, n):
    """BASELINE configs[3] (9 views 928x480, D=256): the 8 sources sharded over n ranks vs the same depth map on one rank."""
    out = {'workload': 'cfg4: 9 views (1 ref + 8 src) 928x480, D=256 (BASELINE configs[3])', 'n_gpus': n,
           'mode': 'views: the 8 source views dealt over the ranks, all-to-all + all-gather inside both AANets'}
    if one is not None:
        out['single_gpu'] = {'ms_per_step': one.get('ms_per_step'), 'value': one.get('value'), 'parity': one.get('parity')}
    else:
        out['single_gpu'] = {'ok': False, 'error': err1}
    if shd is None:
        out.update(ok=False, error=err2)
        return out
    out.update(ok=True, value=shd.get('value'), unit=shd.get('unit'), ms_per_step=shd.get('ms_per_step'),
               scaling='strong', source_views_per_sec=shd.get('source_views_per_sec'), exchange=shd.get('exchange'),
               parity=shd.get('parity'), parallelism=shd['config']['parallelism'], groups=shd['config']['groups'],
               rccl=shd['config'].get('rccl'), rccl_ranks=shd['config'].get('rccl_ranks'))
    if one is not None and one.get('ms_per_step') and shd.get('ms_per_step'):
        speedup = one['ms_per_step'] / shd['ms_per_step']
        out['speedup_vs_single_gpu'] = round(speedup, 3)
        out['fraction_of_linear'] = round(speedup / n, 4)
    return out


# --------------------------------------------------------------------------------------------- CPU baseline

def cpu_baseline(args):
    """The CPU oracle (our restatement of the reference's TF-CPU path; TensorFlow 1.5 cannot be installed) timed on the host
    cores: BASELINE configs[0] (two-view 160x128, D=32) 1 warm-up + 3 runs, median; then the benchmark's OWN workload end
    to end, once, at full size (configs[1] / configs[2]: about 12 s / 55 s on 32 threads of the GPU box's host) -- `value`
    is that measured wall time, nothing composed.  For the larger workloads (cfg4 / cfg5: 5-7 minutes of CPU work) every
    stage runs once at full size and `value` composes them the way the pipeline does (`measured_end_to_end`: false)."""
    import numpy as np
    import torch
    from atvsnet_amd import synthetic, variables
    from oracle import model as OM, nets
    ncpu = os.cpu_count() or 1
    W = {k: torch.from_numpy(v) for k, v in variables.default_store().host.items()}
    model_name = ''
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('model name'):
                    model_name = ln.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    t_all = time.time()
    # how many threads: measured, not asserted -- a mid-size two-view scene (320x256, D=96: 1/8 of configs[1], ~1-2 s) once per
    # candidate count, the fastest count runs everything below
    sweep = {}
    imgs, cams = synthetic.make_inputs(2, 256, 320, 96)
    imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
    for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(n)
        with torch.no_grad():
            OM.run_twoview(imgs, cams, W, 96) if not sweep else None      # first call also warms the allocator / oneDNN caches
            t = time.time()
            OM.run_twoview(imgs, cams, W, 96)
            sweep[n] = round(time.time() - t, 3)
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    # configs[0]: full pipeline, 1 warm-up + 3 runs
    imgs, cams = synthetic.make_inputs(2, 128, 160, 32)
    imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
    runs = []
    with torch.no_grad():
        for i in range(4):
            t = time.time()
            OM.run_twoview(imgs, cams, W, 32)
            runs.append(time.time() - t)
    cfg1 = float(np.median(runs[1:]))
    D = args.depths
    n_src = args.views - 1
    base = {'unit': 'depth-maps/sec', 'cores': threads, 'kind': 'port',
            'thread_sweep_s': {str(k): v for k, v in sorted(sweep.items())},
            'thread_sweep_sample': 'two-view 320x256, D=96 (1/8 of configs[1]) once per thread count; the fastest count is `cores`',
            'kind_detail': '"port" = the CPU oracle (torch-CPU / numpy restatement of the reference\'s TF-1.5 path, oracle/), '
                           'NOT TensorFlow and not a build of the reference -- TensorFlow 1.5 cannot be installed here',
            'host_cpus': ncpu, 'cpu_model': model_name,
            'configs0_s': round(cfg1, 3), 'configs0_runs_s': [round(r, 3) for r in runs]}
    if not args.custom and args.workload in ('cfg2', 'cfg3'):
        imgs, cams = synthetic.make_inputs(args.views, args.height, args.width, D, seed=0)
        imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
        with torch.no_grad():
            t = time.time()
            out = OM.run_twoview(imgs, cams, W, D) if n_src == 1 else OM.run_multiview(imgs, cams, W, D)
            per_map = time.time() - t
        base.update(value=1.0 / per_map, measured_end_to_end=True, workload_s=round(per_map, 2),
                    output_checksum=float(out.double().mean()),
                    sample='CPU restatement (not TensorFlow): configs[0] (160x128, D=32 two-view) 1 warm-up + 3 runs, median %.2f s; '
                           'then ONE depth map of the benchmarked workload end to end (%d views %dx%d, D=%d, the same seeded inputs '
                           'and weights as the GPU run): %.1f s measured on %d threads; value = 1 / that; %.0f s of CPU work in all'
                           % (cfg1, args.views, args.width, args.height, D, per_map, threads, time.time() - t_all))
        return base
    # the larger workloads, stage by stage (= one two-view pipeline), each stage once at full size
    imgs, cams = synthetic.make_inputs(2, args.height, args.width, D)
    imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
    ds, di = OM.depth_start_interval(cams)
    T = {}
    with torch.no_grad():
        t = time.time()
        ref_f = nets.resnet_ds2_spp(imgs[:, 0], W)
        T['tower'] = time.time() - t
        view_f = nets.resnet_ds2_spp(imgs[:, 1], W)
        t = time.time()
        cv = OM.build_cost_volume(ref_f, view_f, cams, D, ds, di, 0, 1)
        T['warp'] = time.time() - t
        t = time.time()
        pv, filt = OM.cost_volume_reasoning(cv, W)
        depth = OM.prob2depth(pv, D, ds, di)
        T['unet'] = time.time() - t
        del cv
        t = time.time()
        OM.TVSNet_refine(depth, depth, pv, filt, imgs, cams, D, ds, di, W, view_i=1)
        T['refine'] = time.time() - t
        T['aam'] = 0.0
        if n_src > 1:
            t = time.time()
            agg = nets.attention_aggregation(torch.stack([filt] * n_src, -1), W, 'attention_aggregate')
            OM.prob2depth_upsample(nets.output_conv(agg, W), D, ds, di)
            T['aam'] = time.time() - t
    twoview = 2 * T['tower'] + 2 * (T['warp'] + T['unet']) + T['refine']
    if n_src > 1:
        per_map = args.views * T['tower'] + n_src * (2 * (T['warp'] + T['unet']) + T['refine']) + 2 * T['aam']
    else:
        per_map = twoview
    base.update(value=1.0 / per_map, measured_end_to_end=False, twoview_fullsize_s=round(twoview, 2),
                stage_s={k: round(v, 2) for k, v in T.items()},
                sample='CPU restatement (not TensorFlow): configs[0] (160x128, D=32 two-view) 1 warm-up + 3 runs, median '
                       '%.2f s; every stage of the pipeline once at FULL size %dx%d, D=%d (tower %.1f s, warp %.1f s, '
                       'U-Net %.1f s, refinement %.1f s, AANet over %d views + head + upsample %.1f s); value = 1 / (%d '
                       'towers + %d x (2 warps + 2 U-Nets + refinement) + 2 AAM) = 1 / %.1f s (composed, not measured end to '
                       'end); %.0f s of CPU work in all'
                       % (cfg1, args.width, args.height, D, T['tower'], T['warp'], T['unet'], T['refine'], n_src,
                          T['aam'], args.views if n_src > 1 else 2, n_src, per_map, time.time() - t_all))
    return base


# --------------------------------------------------------------------------------------------- reporting helpers

def hwmon_dir(device=0):
    """The hwmon directory (socket power, shader clock) of HIP device `device`: the drm card with the device's PCI bus id.
    sysfs only -- no child process is started from the GPU-initialised benchmark process."""
    import ctypes
    import glob
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return None
        bus = buf.value.decode().lower()
    except OSError:
        return None
    for card in glob.glob('/sys/class/drm/card*/device'):
        if os.path.basename(os.path.realpath(card)).lower() == bus:
            hw = sorted(glob.glob(os.path.join(card, 'hwmon', 'hwmon*')))
            return hw[0] if hw else None
    return None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def power_probe(step, seconds=2.0, device=0):
    """Socket power and shader clock while `step` repeats for `seconds` (the amdgpu hwmon files of THIS device sampled from a
    second thread), AFTER the timed region: does the workload run into the package's power management?  (It does: DESIGN.md
    4.1, 6.)  None where the files are not readable."""
    import threading
    import numpy as np
    import torch
    hw = hwmon_dir(device)
    if hw is None or _read_int(os.path.join(hw, 'power1_input')) is None:
        return None
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            pw, clk = _read_int(os.path.join(hw, 'power1_input')), _read_int(os.path.join(hw, 'freq1_input'))
            if pw is not None and clk is not None:
                samples.append((time.perf_counter(), pw / 1e6, clk / 1e6))
            time.sleep(0.02)

    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    try:
        while time.perf_counter() - t0 < seconds:
            for _ in range(5):
                step()
            n += 5
            torch.cuda.synchronize()
    finally:
        dt = time.perf_counter() - t0
        stop[0] = True
        th.join(timeout=2)
    warm = [s_ for s_ in samples if 0.5 < s_[0] - t0 < dt]       # past the ramp
    if not warm:
        return None
    cap = _read_int(os.path.join(hw, 'power1_cap'))
    return {'socket_W': round(float(np.mean([s_[1] for s_ in warm])), 1), 'cap_W': None if cap is None else cap / 1e6,
            'sclk_MHz': int(round(float(np.mean([s_[2] for s_ in warm])))), 'peak_sclk_MHz': 2400, 'samples': len(warm),
            'seconds': round(dt, 2), 'ms_per_step': round(1e3 * dt / n, 3), 'source': 'amdgpu hwmon power1_input / freq1_input',
            'note': 'the timed step repeated for ~2 s after the timed region, hwmon sampled every 20 ms (first 0.5 s dropped): the average '
                    'over ALL kernels of the step.  The split-operand convolution launches alone hold the package AT its cap with the '
                    'shader clock far below the 2.4 GHz the MFMA peak is quoted at (tools_dev/power_probe.py, DESIGN.md 4.1: dominant '
                    'launch 1,370-1,400 W at 1.85-1.90 GHz) -- energy per launch, not issue slots, is what bounds them'}


def pmc_traffic(args, samples):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS kernel as bench.py
    launches it (profiles/round3_pmc_xpair.json: the dominant x-pair launch, 8 volumes per launch, 640x512x192; only valid
    for that form and that kernel).  FETCH_SIZE on gfx950 counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM) and
    is uncalibrated for this kernel's 32-byte pieces: the raw and the doubled figure are both given."""
    if (args.width, args.height, args.depths) != (640, 512, 192):
        return None
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            d = json.load(f)['dominant']
        from atvsnet_amd import ops
        if int(d['volumes_per_launch']) != int(samples) or ('conv_%s_kernel' % ops._xkind()) not in d['kernel']:
            return None
        return {'write': int(d['write_bytes']), 'fetch_raw': int(d['fetch_bytes_raw']), 'fetch_x2': 2 * int(d['fetch_bytes_raw']),
                'source': PMC_FILE, 'measured_in_run': False}
    except Exception:
        return None


def pmc_counters(args, samples):
    """Matrix-pipe occupancy, held clock and VALU instructions per MFMA of the dominant launch from the committed PMC passes
    (same validity conditions as pmc_traffic)."""
    if (args.width, args.height, args.depths) != (640, 512, 192):
        return None
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            d = json.load(f)['dominant']
        from atvsnet_amd import ops
        if int(d['volumes_per_launch']) != int(samples) or ('conv_%s_kernel' % ops._xkind()) not in d['kernel']:
            return None
        return {'mfma_busy': d.get('mfma_busy_fraction_of_simd_cycles'), 'clock_GHz': d.get('clock_GHz'),
                'valu_per_mfma': d.get('valu_per_mfma'), 'kernel': d.get('kernel'),
                'duration_ms': round(d['duration_ns_under_profiler'] / 1e6, 4)}
    except Exception:
        return None


def pmc_kernels():
    """The committed per-kernel PMC table of one profiled depth map (tools_dev/pmc_bench.sh + make_pmc_table.py) or None."""
    try:
        with open(os.path.join(ROOT, PMC_KERNELS_FILE)) as f:
            return json.load(f)
    except Exception:
        return None


def conv3d_busy(args):
    """north_star asks 'conv3d >= 40 % MFMA utilisation': the TIME-WEIGHTED MFMA-busy fraction over ALL 3-D convolution kernels of a
    depth map (not the best kernel's), from the committed PMC table -- computed here from the per-kernel entries when the table
    predates the field."""
    if (args.width, args.height, args.depths, args.views) != (640, 512, 192, 5):
        return None
    d = pmc_kernels()
    if not d:
        return None
    tw = d.get('conv3d_mfma_busy_time_weighted')
    if not tw:
        names = ('conv_xb_kernel', 'conv_c16b_kernel', 'conv3d_b_kernel', 'conv3d_s2b_kernel', 'deconv_up_b_kernel', 'deconv_up_b_sum_kernel', 'aanet_b_kernel')
        t = b = 0.0
        for name, e in d.get('kernels', {}).items():
            if any(k in name for k in names) and e.get('mfma_busy') is not None and e.get('total_ms') and (e.get('clock_GHz') or 0) <= 2.4:
                t += e['total_ms']
                b += e['total_ms'] * e['mfma_busy']
        tw = {'value': round(b / t, 3) if t else None, 'over_ms_of_conv3d_kernels': round(t, 3), 'kernels': list(names)}
    return dict(tw, source=PMC_KERNELS_FILE, head=profile_head(PMC_KERNELS_FILE), measured_in_run=False, target=0.40)


def warp_traffic(args):
    """Memory-side bytes per launch of the cost-volume warp from the committed PMC table (FETCH_SIZE / WRITE_SIZE passes)."""
    if (args.width, args.height, args.depths) != (640, 512, 192):
        return None
    d = pmc_kernels()
    e = (d or {}).get('kernels', {}).get('warp_planes_shared_kernel<0, true>')
    if not e:
        return None
    wr, rd = int(e['write_MB_raw'] * 1048576), int(e['fetch_MB_raw'] * 1048576)
    return {'write': wr, 'fetch_raw': rd, 'source': PMC_KERNELS_FILE, 'head': profile_head(PMC_KERNELS_FILE), 'measured_in_run': False,
            'note': 'WRITE_SIZE + FETCH_SIZE of this kernel per launch (KB at the memory side of L2 -> bytes); its reads are 16-byte '
                    'gathers from the 2.6 MB source map (L2 hits never reach the counter), so the x2 correction of wide streaming '
                    'reads does not apply'}


def top_kernels(k=5):
    """Top kernels of the committed rocprofv3 --kernel-trace --stats summary of this command with --inflight 1 (one
    depth map at a time: the kernel times add up to the step; with two in flight -- profiles/
    round2_bench_inflight2_kernel_stats.csv -- small kernels wait behind the other stream's and their durations
    overlap; the dominant kernel's average is the same in both: 5.97 ms)."""
    import csv
    path = os.path.join(ROOT, KERNEL_STATS_FILE)
    if not os.path.exists(path):
        return None
    rows = []
    with open(path) as f:
        for r in csv.DictReader(ln for ln in f if not ln.startswith('#')):
            rows.append((r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0],
                         int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    rows.sort(key=lambda t: -t[3])
    return {'source': KERNEL_STATS_FILE, 'head': profile_head(KERNEL_STATS_FILE), 'measured_in_run': False, 'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --inflight 1',
            'top': [{'kernel': n, 'calls': c, 'avg_us': round(a, 1), 'pct': round(p, 2)} for n, c, a, p in rows[:k]]}


def profiled_avg_us(substr):
    """Average duration (us) of the kernels whose name contains `substr` in the committed kernel-trace summary."""
    import csv
    path = os.path.join(ROOT, KERNEL_STATS_FILE)
    if not os.path.exists(path):
        return None
    tot = calls = 0
    with open(path) as f:
        for r in csv.DictReader(ln for ln in f if not ln.startswith('#')):
            if substr in r['Name']:
                tot += float(r['AverageNs']) * int(r['Calls'])
                calls += int(r['Calls'])
    return tot / calls / 1e3 if calls else None


def parity_check(args, out, eager_out):
    """rel-L1 of the timed path's depth map against the oracle fixture of this workload (when the run is one of
    BASELINE's configurations) + graph replay == eager launch, bit for bit."""
    import numpy as np
    import torch
    res = {'bar': 1e-3}
    if eager_out is not None:
        res['graph_equals_eager_bitwise'] = bool(torch.equal(out, eager_out))
    path = os.path.join(ROOT, 'tests', 'golden', 'fullsize_%s.npz' % args.workload)
    if args.custom or not os.path.exists(path):
        res['fixture'] = None
        return res
    want = torch.from_numpy(np.load(path)['depth'])
    got = out.reshape(want.shape).cpu()
    res['fixture'] = os.path.relpath(path, ROOT)
    res['rel_l1'] = float(((got - want).abs() / want.abs()).mean())
    res['rel_l1_inverse'] = float(((1.0 / got - 1.0 / want).abs() / (1.0 / want).abs()).mean())
    res['max_abs'] = float((got - want).abs().max())
    res['ok'] = bool(res['rel_l1'] <= res['bar'] and res['rel_l1_inverse'] <= res['bar'])
    return res


# --------------------------------------------------------------------------------------------- one rank

def dry_rank(args):
    """Launcher / rendezvous check without a GPU: gloo process group, the plan, one all-reduce."""
    import torch
    import torch.distributed as dist
    import atvsnet_amd                                       # noqa: F401
    from atvsnet_amd import parallel
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    assert world == args.gpus, 'WORLD_SIZE %d != --gpus %d' % (world, args.gpus)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    groups = parallel.rank_groups(args.views, world, args.split_directions)
    t = torch.ones(1)
    if world > 1:
        dist.all_reduce(t)
        dist.destroy_process_group()
    assert int(t.item()) == world
    if rank == 0:
        print(json.dumps({'metric': 'depth-maps/sec at %dx%dxD=%d, N=%d views' % (args.width, args.height, args.depths, args.views),
                          'value': 0.0, 'unit': 'depth-maps/sec', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'dry': True,
                          'config': {'workload': args.workload, 'world_size': world, 'groups': groups,
                                     'plan': parallel.plan(args.views, len(groups[0]))}}))
    return 0


def rank_main(args):
    failed = False
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: WORLD_SIZE=%d but --gpus %d\n' % (world, args.gpus))
        return 2
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    assert torch.cuda.device_count() > local_rank, 'rank %d: no device %d' % (rank, local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)

    import atvsnet_amd                                       # noqa: F401
    from atvsnet_amd import ops, synthetic, variables
    from atvsnet_amd.atvsnet import example as ex
    from atvsnet_amd import parallel

    variables.default_store().init_synthetic(1234)
    imgs, cams = synthetic.make_inputs(args.views, args.height, args.width, args.depths, seed=0)
    imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
    twoview = args.views == 2

    # maps: every rank is its own group (its own depth maps, no data-path collective); views: groups of at most one rank
    # per source view, the sources of one depth map sharded inside a group
    mode = args.parallel if args.parallel in ('maps', 'views') else 'maps'
    if world > 1 and (mode == 'maps' or twoview):
        mode, groups = 'maps', [[r] for r in range(world)]
    else:
        groups = parallel.rank_groups(args.views, world, args.split_directions) if world > 1 else [[0]]
    group, gsize, n_groups = None, 1, len(groups)
    if world > 1:
        for g in groups:
            h = dist.new_group(g) if (len(groups) > 1 and len(g) > 1) else None       # every rank creates every group
            if rank in g:
                group, gsize = h, len(g)
    sharded = gsize > 1

    graphed, pipe = None, None
    if not args.eager:
        if sharded:
            # per rank: a chain of HIP graphs (the local compute between two exchanges) with the RCCL calls between
            graphed = parallel.ShardedGraphedInference(imgs, cams, args.depths, group=group)
        else:
            # one HIP graph per depth map; `inflight` of them (own static buffers and stream each) for the timed region
            pipe = ex.PipelinedInference(imgs, cams, args.depths, slots=max(1, args.inflight), co_resident=True)
            graphed = pipe.graphs[0]

    def eager_step(view_streams=True):
        if sharded:
            return parallel.infer_multiview_sharded(imgs, cams, args.depths, group=group, view_streams=view_streams)
        if twoview:
            return ex.infer_twoview(imgs, cams, args.depths)
        return ex.infer_multiview(imgs, cams, args.depths, view_streams=view_streams)

    def step():
        return graphed() if graphed is not None else eager_step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region = K depth maps strictly one after the other (the metric is 1 / wall time of ONE depth map,
    # images resident -> depth map resident).  The pipelined rate (`inflight` maps in flight on their own streams) is
    # measured in a second region and reported under `pipelined`, never as `value`.
    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    pipelined = None
    if pipe is not None and pipe.slots > 1:
        pipe.run(args.warmup)
        barrier()
        t1 = time.perf_counter()
        pipe.run(args.steps)                 # K depth maps, round-robin over the slots' streams
        barrier()
        dtp = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dtp], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtp = float(t.item())
        # every slot computed the captured inputs: all of them must hold the depth map the timed single-map path produced
        same = all(bool(torch.equal(g.out, out)) for g in pipe.graphs)
        pipelined = {'value': round(n_groups * args.steps / dtp, 4), 'unit': 'depth-maps/sec', 'inflight': pipe.slots,
                     'ms_per_step': round(1e3 * dtp / args.steps, 3), 'equals_single_map_bitwise': same, 'co_resident': True,
                     'note': '%d depth maps in flight with their kernels co-resident on the GPU (example.PipelinedInference('
                             'co_resident=True): opt-in, the product default runs one map at a time -- DESIGN.md appendix B): '
                             'throughput of independent depth maps of a scene, NOT the per-map rate `value` reports' % pipe.slots}
    pipelined_cu_split = None
    if pipe is not None and pipe.slots > 1:
        # the same `inflight` maps in flight, each on its own share of every XCD's compute units (example.cu_split_streams): no SIMD
        # is shared by two kernels, so the co-residency fault cannot occur -- the SAFE form of maps in flight
        try:
            pipe.set_mode('cu_split')
            # a whole number of rounds over the slots and at least 12 maps: a map confined to 1 / slots of the chip takes longer than
            # one on the whole chip, so an odd last map (K = 5 on two slots) would measure the tail, not the rate
            ksplit = -(-max(args.steps, 12) // pipe.slots) * pipe.slots
            pipe.run(pipe.slots * max(1, -(-args.warmup // pipe.slots)))
            barrier()
            t1 = time.perf_counter()
            pipe.run(ksplit)
            barrier()
            dts = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([dts], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dts = float(t.item())
            same = all(bool(torch.equal(g.out, out)) for g in pipe.graphs)
            pipelined_cu_split = {'value': round(n_groups * ksplit / dts, 4), 'unit': 'depth-maps/sec', 'inflight': pipe.slots,
                                  'maps_timed': ksplit, 'ms_per_step': round(1e3 * dts / ksplit, 3), 'equals_single_map_bitwise': same,
                                  'note': '%d depth maps in flight, each confined to its own 1/%d of every XCD (hipExtStreamCreateWithCUMask, '
                                          "example.PipelinedInference(co_resident='cu_split')): kernels of different maps never share a "
                                          'SIMD; throughput of independent depth maps, NOT the per-map rate `value` reports' % (pipe.slots, pipe.slots)}
        except Exception as e:                 # a runtime without the CU-mask extension: reported, the primary number stands
            pipelined_cu_split = {'error': repr(e)}
        finally:
            try:
                pipe.set_mode(True)
            except Exception:
                pass
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = out.clone()

    # secondary ("also report 5-source N = 6"): the metric's configuration with the reference's DEFAULT view_num = 5
    # SOURCE views + the reference view (flags.view_num = 5 counts the images of example/0,1: 5 images = 4 sources -- the metric's
    # N = 5; a scene with five sources is N = 6): one more pass of every per-view network, AANet modules over five views in one launch
    five = None
    if world == 1 and graphed is not None and not args.custom and args.workload == 'cfg3' and not args.no_fp32_path:
        try:
            im6, cm6 = synthetic.make_inputs(6, args.height, args.width, args.depths, seed=0)
            im6, cm6 = torch.from_numpy(im6).to(dev), torch.from_numpy(cm6).to(dev)
            g6 = ex.GraphedInference(im6, cm6, args.depths)
            e6 = ex.infer_multiview(im6, cm6, args.depths)
            for _ in range(args.warmup):
                g6()
            torch.cuda.synchronize()
            t6 = time.perf_counter()
            for _ in range(args.steps):
                o6 = g6()
            torch.cuda.synchronize()
            dt6 = time.perf_counter() - t6
            five = {'config': {'workload': '1 depth map per step: 6 views (1 ref + 5 src) %dx%d, D=%d' % (args.width, args.height, args.depths)},
                    'value': round(args.steps / dt6, 4), 'unit': 'depth-maps/sec', 'ms_per_step': round(1e3 * dt6 / args.steps, 3),
                    'source_views_per_sec': round(5 * args.steps / dt6, 3), 'graph_equals_eager_bitwise': bool(torch.equal(o6, e6)),
                    'finite': bool(torch.isfinite(o6).all())}
            fx = os.path.join(ROOT, 'tests', 'golden', 'fullsize_cfg3s5.npz')
            if os.path.exists(fx) and not args.no_parity:
                want6 = torch.from_numpy(np.load(fx)['depth'])
                got6 = o6.reshape(want6.shape).cpu()
                five['parity'] = {'fixture': os.path.relpath(fx, ROOT), 'bar': 1e-3,
                                  'rel_l1': float(((got6 - want6).abs() / want6.abs()).mean()),
                                  'rel_l1_inverse': float(((1.0 / got6 - 1.0 / want6).abs() / (1.0 / want6).abs()).mean())}
                five['parity']['ok'] = bool(five['parity']['rel_l1'] <= 1e-3 and five['parity']['rel_l1_inverse'] <= 1e-3)
            del g6, e6, o6, im6, cm6
        except Exception as e:
            five = {'error': repr(e)}

    # secondary: the same step with EVERY convolution on the fp32 matrix cores (`ops.configure(split16=False)`).  The default path runs
    # its heavy layers on v_mfma_f32_16x16x32_f16 with SPLIT operands: x = h0 + h1 / 2048, w = g0 + g1 / 2048 in fp16 (22
    # significant bits), the three products h0 g0 + (h0 g1 + h1 g0) / 2048, fp32 accumulation: fp32-class results
    # (per-layer error against float64 below the fp32 MFMA kernel's; BASELINE configs[1] names "bf16 conv3d MFMA") -- so the
    # all-fp32-MFMA figure is printed next to it.
    split = None
    if world == 1 and graphed is not None and not args.no_fp32_path:
        try:
            default_on = ops.cfg.split16
            ops.cfg.split16 = not default_on
            g2 = ex.GraphedInference(imgs, cams, args.depths)
            for _ in range(args.warmup):
                g2()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                o2 = g2()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            split = {'layers': 'the 3x3x3 layers with 8 / 16 / 32 input channels (conv_xb.hip, conv_c16b.hip) and the 3x3 / 1x1 tower '
                               'layers with Cin % 32 == 0 (conv2d_b.hip, conv1x1_b.hip) on v_mfma_f32_16x16x32_f16: '
                               'x = h0 + h1 / 2048, w = g0 + g1 / 2048 in fp16 (h1 = f16((x - h0) * 2048)), the 3 products '
                               'h0 g0 + (h0 g1 + h1 g0) / 2048, fp32 accumulation; the transposed convolutions (deconv_up_b.hip) likewise',
                     'in_value': bool(default_on),
                     'other_path': 'every convolution on the fp32 matrix cores' if default_on else 'split-operand layers enabled',
                     'other_ms_per_step': round(1e3 * dt2 / args.steps, 3), 'other_value': round(args.steps / dt2, 4),
                     'unit': 'depth-maps/sec', 'other_parity': None if args.no_parity else parity_check(args, o2.clone(), None)}
            del g2
        except Exception as e:
            split = {'error': repr(e)}
        finally:
            ops.cfg.split16 = default_on

    power = None
    if world == 1 and not args.no_power:
        try:
            power = power_probe(step)
        except Exception as e:                                # noqa: BLE001
            power = {'error': repr(e)}

    # the dominant kernel and the warp, timed by HIP events on their launch stream in ten eager single-stream passes
    # of the same step right after the timed region (kernels inside a replayed graph cannot be bracketed by events;
    # single stream: no other kernel shares the GPU with the one being timed)
    ops.watch([DOMINANT, WARP])
    for _ in range(EAGER_TIMING_PASSES):
        eager_out = eager_step(view_streams=False)
    watched = ops.watch(None)
    assert torch.isfinite(out).all()
    comm = None
    if sharded and graphed is not None and hasattr(graphed, 'timed_call'):
        comm = graphed.timed_call()              # ms of local graphs vs ms of the exchanges, this rank
    if world > 1:
        dist.barrier()

    if rank == 0:
        h, w = args.height // 4, args.width // 4
        vox = args.depths * h * w
        # conv_b0_0_1 (3x3x3, 64 -> 8, stride 1, SAME) and conv_b0_1_0 (3x3x3, 64 -> 16, stride 2, SAME) read the same
        # cost volume and run as ONE launch.  Their 32 D-constant input channels (the tiled reference features) are
        # per-plane biases, so the launch convolves the 32 warped channels: 2*27*32*8 FLOP per voxel + 2*27*32*16 FLOP
        # per half-resolution voxel.  Without the sibling (`ops.configure(siblings=False)`) only the first term applies.
        vox2 = ((args.depths + 1) // 2) * ((h + 1) // 2) * ((w + 1) // 2)
        sib = ops.siblings_ok((args.depths, h, w), 32, 8, 16)
        flops = 2.0 * 27 * 32 * 8 * vox + (2.0 * 27 * 32 * 16 * vox2 if sib else 0.0)
        alg_bytes = 4 * (32 * vox + 8 * vox + (16 * vox2 if sib else 0))
        roof = roof_hbm = None
        if watched.get(DOMINANT):
            # a launch covers `samples` independent (view, direction) volumes (batched evaluation): the per-volume
            # figures above times the samples of the launch, over the launch's duration
            samples = float(np.mean([g for _, g in watched[DOMINANT]]))
            avg_ms = float(np.mean([ms for ms, _ in watched[DOMINANT]]))
            flops, alg_bytes = flops * samples, int(alg_bytes * samples)
            ach = flops / (avg_ms * 1e-3) / 1e12
            tr = pmc_traffic(args, samples)
            kind = ops._xkind()
            kdesc = {'xb': 'conv_xb_kernel<SIB> (x-pair rows on v_mfma_f32_16x16x32_f16, every fp32 operand split into two fp16 '
                           'pieces, three products, fp32 accumulation)',
                     'xw': 'conv_xw_kernel<SIB> (x-pair rows x Winograd F(2,3) along y on v_mfma_f32_16x16x4_f32)'}[kind]
            pmc = pmc_counters(args, samples)
            fp32_eq = {'achieved': round(ach, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                       'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                       'note': 'ALGORITHMIC direct fp32 convolution FLOPs (SURVEY 8d) / time against the fp32 matrix peak the path is '
                               'specified in: what an fp32-MFMA kernel would need to reach; exceeds 1.0 on the 16-bit pipe '
                               '(ceiling 2500 / 4 / 157.3 = 3.97), so it is NOT the roofline fraction'}
            if kind == 'xb':
                # main: 9 steps x 3 products per (8-channel chunk, 16 pairs x row): 2*16*16*32 FLOP each = 4 x the useful
                # fp32 FLOPs (x-pair zero taps 4/3, three products); sibling: 7 steps of 4 taps for 27 -> 28/27 x 3
                main = 2.0 * 27 * 32 * 8 * vox * samples
                sibf = (flops - main)
                issued = main * 4.0 + sibf * 3.0 * 28.0 / 27.0
                peak, pipe = PEAK_BF16_MFMA_TFLOPS, 'fp16'
                conv = ('fp16 MFMA FLOPs the launch ISSUES (v_mfma_f32_16x16x32_f16 count x 16384; 4 per algorithmic '
                        'FLOP of the main convolution: three piece products x 4/3 x-pair rows; 3 x 28/27 for the sibling) / '
                        'time, against the dense fp16 / bf16 MFMA peak: the pipe the kernel runs on')
            else:
                issued, peak, pipe = flops * 8.0 / 9.0, PEAK_F32_MFMA_TFLOPS, 'fp32'
                conv = 'fp32 MFMA FLOPs issued (F(2,3): 2/3, x-pair rows: 4/3 of the algorithmic FLOPs) / time'
            iss = issued / (avg_ms * 1e-3) / 1e12
            # achieved = ALGORITHMIC convolution FLOPs of the launch / its duration, against the dense 16-bit MFMA
            # peak of the pipe it runs on.  What the pipe is busy with (three piece products, x-pair zero rows) is NOT in
            # `frac`: it is reported apart as `mfma_pipe_utilisation` (a kernel that issued redundant MFMAs would score higher
            # there, never here).  `emulation_ceiling_frac`: against peak / 3 -- the most an fp32-accurate three-product scheme
            # can reach on this pipe.
            products = 3.0 if kind == 'xb' else 1.0
            roof = {'bound': 'mfma', 'pipe': pipe,
                    'kernel': '%s: conv_b0_0_1 (32 warped channels -> 8, 3x3x3, full resolution) + sibling conv_b0_1_0 '
                              '(-> 16, stride 2), %d volumes per launch' % (kdesc, int(samples)),
                    'flops_convention': 'achieved = algorithmic direct-convolution FLOPs of the launch (2*27*Cin*Cout per output voxel, '
                                        'SURVEY 8d) / its duration from HIP events in this run; peak = dense fp16 / bf16 MFMA',
                    'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                    'algorithmic_frac': round(ach / peak, 4),
                    'emulation_ceiling_frac': round(ach * products / peak, 4),
                    'emulation_ceiling_TFLOPs': round(peak / products, 1),
                    'mfma_pipe_utilisation': {'achieved': round(iss, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(iss / peak, 4),
                                              'issued_flops_per_launch': issued, 'convention': conv,
                                              'note': 'ISSUED MFMA work, not useful work: a utilisation of the pipe, not a roofline fraction'},
                    'fp32_equivalent': fp32_eq,
                    'mfma_busy': pmc.get('mfma_busy') if pmc else None,
                    'conv3d_mfma_busy_time_weighted': conv3d_busy(args),
                    'clock_GHz': pmc.get('clock_GHz') if pmc else None,
                    'valu_per_mfma': pmc.get('valu_per_mfma') if pmc else None,
                    'pmc': ({'source': PMC_FILE, 'head': profile_head(PMC_FILE), 'measured_in_run': False, 'kernel': pmc.get('kernel'),
                             'duration_ms_under_profiler': pmc.get('duration_ms'),
                             'note': 'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs); cycles = min(GRBM_GUI_ACTIVE / 8, '
                                     'SQ_BUSY_CYCLES / 32) (tools_dev/pmc_derive.py); clock_GHz = cycles / duration: the clock the chip '
                                     'holds under this kernel (2.4 GHz spec)'}
                            if pmc else None),
                    'traffic': (tr['write'] + tr['fetch_x2']) if tr else None,
                    'traffic_detail': dict(tr, algorithmic=alg_bytes,
                                           ratio_raw=round((tr['write'] + tr['fetch_raw']) / alg_bytes, 3),
                                           ratio_x2=round((tr['write'] + tr['fetch_x2']) / alg_bytes, 3),
                                           note='memory-side bytes per launch from rocprofv3 --pmc passes of this kernel (8 volumes '
                                                'per launch); FETCH_SIZE counts Infinity-Cache hits too and on gfx950 reports half '
                                                'the bytes of wide coalesced reads: the true read traffic lies between fetch_raw '
                                                'and fetch_x2; `traffic` uses write + fetch_x2')
                    if tr else None,
                    'avg_launch_ms': round(avg_ms, 4), 'launches': len(watched[DOMINANT]), 'volumes_per_launch': samples,
                    'algorithmic_flops_per_launch': flops, 'algorithmic_bytes_per_launch': alg_bytes}
        if watched.get(WARP):
            # plane-sweep warp of the 32-channel source features into the D-varying half of the cost volume:
            # algorithmic bytes = write D*h*w*32*4 + read h*w*32*4
            wb = 4.0 * 32 * vox + 4.0 * 32 * h * w
            avg_ms = float(np.mean([ms for ms, _ in watched[WARP]]))
            gbs = wb / (avg_ms * 1e-3) / 1e9
            roof_hbm = {'bound': 'hbm', 'kernel': 'warp_planes_shared_kernel<bilinear> (cost-volume build, 32 channels; geometry once per pixel)',
                        'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4),
                        'timing': 'HIP events around eager launches in this run (includes launch gaps of ~10 us)',
                        'avg_launch_ms': round(avg_ms, 4), 'launches': len(watched[WARP]),
                        'algorithmic_bytes_per_launch': wb, 'traffic': None}
            wt = warp_traffic(args)
            if wt:
                roof_hbm['traffic'] = wt['write'] + wt['fetch_raw']
                roof_hbm['traffic_detail'] = dict(wt, ratio=round((wt['write'] + wt['fetch_raw']) / wb, 3))
            prof_us = profiled_avg_us('warp_planes_shared_kernel<0,')
            if prof_us and (args.width, args.height, args.depths) == (640, 512, 192):
                roof_hbm['profile'] = {'source': KERNEL_STATS_FILE, 'head': profile_head(KERNEL_STATS_FILE), 'measured_in_run': False, 'avg_launch_ms': round(prof_us / 1e3, 4),
                                       'achieved': round(wb / (prof_us * 1e-6) / 1e9, 1),
                                       'frac': round(wb / (prof_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}
        par = 'single GPU'
        if world > 1 and not sharded:
            par = '%d ranks, one depth map each per step (independent reference views of a scene), no data-path collective' % world
        elif world > 1:
            par = ('%d group(s) of %d rank(s); inside a group the source views of one depth map are sharded over the ranks, '
                   'exchange inside AAM1/AAM2 over RCCL (%s)' % (n_groups, len(groups[0]), parallel.EXCHANGE))
        line = {
            'metric': 'depth-maps/sec at %dx%dxD=%d, N=%d views' % (args.width, args.height, args.depths, args.views),
            'value': round(n_groups * args.steps / dt, 4),
            'unit': 'depth-maps/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True,
            'scaling': 'strong' if sharded else 'weak', 'vs_baseline': None,
            'dtype': 'f32 (fp16x2 split operands)' if ops.cfg.split16 else 'f32', 'dtype_operands': 'two fp16 pieces per fp32 operand, three '
            'MFMA products, fp32 accumulation (22-bit operands, fp16 range; fp32 rerun on overflow)' if ops.cfg.split16 else 'fp32',
            'data': 'synthetic',
            'precision': ('fp32 storage and fp32 accumulation everywhere; the heavy convolutions (3x3x3 with 8 / 16 / 32 input '
                          'channels, 3x3 and 1x1 tower layers) split every fp32 operand into two fp16 pieces (22 significant bits, '
                          'the residual piece scaled into the normal range) '
                          'and form 3 products with the cross terms accumulated apart: fp32-class (per-layer error against float64 '
                          'below the fp32 matrix cores\'), same parity bar; the other layers use fp32 MFMA operands; see '
                          '`split_operands` for the all-fp32-MFMA figure') if ops.cfg.split16 else 'fp32 MFMA operands, fp32 accumulation',
            'config': {'workload': '%d depth map(s) per step: %d views (1 ref + %d src) %dx%d, D=%d, example.py %s pipeline'
                                   % (n_groups, args.views, args.views - 1, args.width, args.height, args.depths,
                                      'two-view' if twoview else 'multi-view'),
                       'name': args.workload if not args.custom else 'custom',
                       'feature_hw': [h, w], 'voxels': vox, 'parallelism': par, 'world_size': world,
                       'groups': groups if world > 1 else None,
                       'rccl': '.'.join(str(v) for v in torch.cuda.nccl.version()) if world > 1 else None,
                       # ranks of the communicator the DATA PATH uses: 0 in maps mode (no data-path collective)
                       'rccl_ranks': gsize if sharded else 0,
                       'launch': 'eager' if graphed is None else ('HIP graph replay, batched per-view networks' if not sharded else
                                                                   'HIP graphs between the exchanges'),
                       'inflight': 1},
            'source_views_per_sec': round(n_groups * args.steps * (args.views - 1) / dt, 3),
            'latency_ms': round(1e3 * dt / args.steps, 3),
            'pipelined': pipelined,
            'pipelined_cu_split': pipelined_cu_split,
            'split_operands': split,
            'five_sources': five,
            'roofline': roof, 'roofline_hbm': roof_hbm, 'power': power, 'kernels': top_kernels(),
        }
        if comm is not None:
            line['exchange'] = comm
        if not args.no_parity:
            try:
                line['parity'] = parity_check(args, out, eager_out if (graphed is not None and not sharded) else None)
            except Exception as e:
                line['parity'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(args)
            except Exception as e:                        # the baseline must never hide the GPU number
                line['cpu_baseline'] = {'error': repr(e)}
        problems = []
        if pipelined is not None and not pipelined['equals_single_map_bitwise']:
            problems.append('pipelined: a slot of the co-resident run differs from the single-map output')
        if pipelined_cu_split is not None and pipelined_cu_split.get('equals_single_map_bitwise') is False:
            problems.append('pipelined_cu_split: a slot of the CU-split run differs from the single-map output')
        if isinstance(line.get('parity'), dict) and line['parity'].get('ok') is False:
            problems.append('parity: rel-L1 above the bar')
        if isinstance(line.get('parity'), dict) and line['parity'].get('graph_equals_eager_bitwise') is False:
            problems.append('parity: graph replay differs from eager launch')
        line['ok'] = not problems
        if problems:
            line['problems'] = problems
            failed = True
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    return 4 if failed else 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        sys.stderr.write('bench.py: --gpus must be >= 1\n')
        return 2
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch(args, argv)
    if args.dry:
        return dry_rank(args)
    return rank_main(args)


if __name__ == '__main__':
    sys.exit(main())
