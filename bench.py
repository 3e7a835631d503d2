#!/usr/bin/env python
"""Headline benchmark: depth-maps/sec of the A-TVSNet multi-view inference path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one depth map: N=5 views (1 reference + 4 sources) of 640x512 images, D=192
hypotheses (BASELINE.json configs[2], the configuration the metric is quoted on), through the
whole example.py multi-view pipeline (towers -> 2x stacked 3-D U-Net per source -> AAM1 ->
refinement per source -> AAM2 -> x4 upsample + soft-argmin), inputs resident in HBM, synthetic
seeded data and weights.  Arithmetic: fp32 (fp32 MFMA for every convolution).
On one GPU the step is ONE replay of a HIP graph captured from the pipeline (per-view streams
forked and joined inside it); --eager issues every launch from Python instead.

--gpus N > 1 (one process per GPU, launched by torch.distributed.run): the source views of
the SAME depth map are sharded over the ranks and aggregated with RCCL all-reduces inside both
AANet modules (a-tvsnet_amd/parallel.py); per rank the local compute between two all-reduces is one
HIP graph; total work is fixed -> "scaling": "strong".

One JSON line on rank 0; `roofline` is for the dominant kernel (conv_xp.hip: the 3x3x3 convolution of
the 32 warped channels of conv_b0_0_1 at full resolution together with its stride-2 sibling conv_b0_1_0,
one launch), timed with HIP events on its launch stream; `roofline.traffic` is the HBM traffic of that
launch from rocprofv3 PMC passes recorded in profiles/round1_pmc_dominant_kernel_v2.json (WRITE_SIZE +
FETCH_SIZE, see the note there);
`cpu_baseline` is the CPU oracle on the host cores over a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import torch                # noqa: E402

WIDTH, HEIGHT, DEPTHS, VIEWS = 640, 512, 192, 5
DOMINANT = ('conv_b0_0_1/conv3d/kernel', 'var')   # the D-varying half of conv_b0_0_1 (see ops.conv_split)
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md, Peak FP32 (matrix)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--width', type=int, default=WIDTH)
    p.add_argument('--height', type=int, default=HEIGHT)
    p.add_argument('--depths', type=int, default=DEPTHS)
    p.add_argument('--views', type=int, default=VIEWS)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--eager', action='store_true', help='issue every launch from Python instead of replaying a HIP graph')
    p.add_argument('--cpu-baseline-seconds', type=float, default=25.0)
    return p.parse_args()


def cpu_baseline(args, budget_s):
    """Time the CPU oracle (our restatement of the reference's TF-CPU path; TF-1.5 cannot be
    installed) on the host cores over a BOUNDED sample of the same workload: every stage of the
    pipeline once on a (W/2 x H/2 image, D/2 hypotheses) crop, i.e. 1/4 of the pixels of a tower
    pass and 1/8 of the voxels of a volume pass, scaled back linearly (all stages are
    convolutions / gathers, linear in pixels resp. voxels) and composed as the pipeline composes
    them."""
    from atvsnet_amd import synthetic, variables
    from oracle import model as OM, nets
    threads = min(os.cpu_count() or 1, 32)      # oneDNN scales badly past a few dozen threads on these sizes
    torch.set_num_threads(threads)
    W = {k: torch.from_numpy(v) for k, v in variables.default_store().host.items()}
    Hs, Ws, Ds = args.height // 2, args.width // 2, max(args.depths // 2, 8)
    pix_ratio = (args.height * args.width) / float(Hs * Ws)
    vox_ratio = pix_ratio * args.depths / float(Ds)
    n_src = args.views - 1
    t0 = time.time()

    def stages(H, Wd, D, timed):
        imgs, cams = synthetic.make_inputs(2, H, Wd, D)
        imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
        ds, di = OM.depth_start_interval(cams)
        T = {}
        with torch.no_grad():
            t = time.time()
            ref_f = nets.resnet_ds2_spp(imgs[:, 0], W)
            T['tower'] = time.time() - t
            view_f = nets.resnet_ds2_spp(imgs[:, 1], W) if not timed else ref_f.flip(2).contiguous()
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
            t = time.time()
            agg = nets.attention_aggregation(torch.stack([filt] * n_src, -1), W, 'attention_aggregate')
            OM.prob2depth_upsample(nets.output_conv(agg, W), D, ds, di)
            T['aam'] = time.time() - t
        return T

    stages(128, 160, 8, False)                   # warm-up: oneDNN primitive creation, allocator
    T = stages(Hs, Ws, Ds, True)
    per_map = (args.views * T['tower'] * pix_ratio +
               n_src * (2 * (T['warp'] + T['unet']) + T['refine']) * vox_ratio + 2 * T['aam'] * vox_ratio)
    return {'value': 1.0 / per_map, 'unit': 'depth-maps/sec', 'cores': threads, 'kind': 'port',
            'sample': 'each stage once on a %dx%d, D=%d crop (tower %.2fs, warp %.2fs, U-Net %.2fs, refinement %.2fs, '
                      'AAM %.2fs), scaled x%.0f (pixels) / x%.0f (voxels) and composed as %d towers + %d x (2 warps + '
                      '2 U-Nets + refinement) + 2 AAM; %.1f s of CPU work measured'
                      % (Ws, Hs, Ds, T['tower'], T['warp'], T['unet'], T['refine'], T['aam'], pix_ratio, vox_ratio,
                         args.views, n_src, time.time() - t0)}


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (only valid for
    the default workload they were taken on): WRITE_SIZE + raw FETCH_SIZE."""
    if (args.width, args.height, args.depths) != (WIDTH, HEIGHT, DEPTHS):
        return None
    try:
        with open(os.path.join(ROOT, 'profiles', 'round1_pmc_dominant_kernel_v2.json')) as f:
            d = json.load(f)['derived']
        return d['write_bytes_pmc'] + d['fetch_bytes_pmc_raw']      # bytes per launch
    except Exception:
        return None


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)

    import atvsnet_amd                                       # noqa: F401
    from atvsnet_amd import ops, synthetic, variables
    from atvsnet_amd.atvsnet import example as ex
    from atvsnet_amd import parallel

    variables.default_store().init_synthetic(1234)
    imgs, cams = synthetic.make_inputs(args.views, args.height, args.width, args.depths, seed=0)
    imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)

    graphed = None
    if not args.eager:
        if world == 1:
            graphed = ex.GraphedInference(imgs, cams, args.depths)      # one HIP graph per depth map
        else:
            # per rank: a chain of HIP graphs (the local compute between two all-reduces) with the RCCL calls between
            graphed = parallel.ShardedGraphedInference(imgs, cams, args.depths)

    def eager_step():
        if world > 1:
            return parallel.infer_multiview_sharded(imgs, cams, args.depths)
        if args.views == 2:
            return ex.infer_twoview(imgs, cams, args.depths)
        return ex.infer_multiview(imgs, cams, args.depths)

    def step():
        return graphed() if graphed is not None else eager_step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    if graphed is None:
        ops.watch(DOMINANT)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if graphed is None:
        watched = ops.watch(None)
    else:
        # kernels inside a replayed graph cannot be bracketed by events: the dominant kernel is timed by
        # HIP events on its launch stream in two eager single-stream passes of the same step (no other kernel
        # shares the GPU with it), right after the timed region
        ops.watch(DOMINANT)
        for _ in range(2):
            if world > 1:
                parallel.infer_multiview_sharded(imgs, cams, args.depths, view_streams=False)
            elif args.views > 2:
                ex.infer_multiview(imgs, cams, args.depths, view_streams=False)
            else:
                eager_step()
        watched = ops.watch(None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        h, w = args.height // 4, args.width // 4
        vox = args.depths * h * w
        # conv_b0_0_1 (3x3x3, 64 -> 8, stride 1, SAME) and conv_b0_1_0 (3x3x3, 64 -> 16, stride 2, SAME) read the same
        # cost volume and run as ONE launch.  Their 32 D-constant input channels (the tiled reference features) are
        # per-plane biases, so the launch convolves the 32 warped channels: 2*27*32*8 FLOP per voxel + 2*27*32*16 FLOP
        # per half-resolution voxel.  Without the sibling (ops.use_siblings(False)) only the first term applies.
        from atvsnet_amd import ops as _ops
        vox2 = ((args.depths + 1) // 2) * ((h + 1) // 2) * ((w + 1) // 2)
        sib = _ops.siblings_ok((args.depths, h, w), 32, 8, 16)
        flops = 2.0 * 27 * 32 * 8 * vox + (2.0 * 27 * 32 * 16 * vox2 if sib else 0.0)
        alg_bytes = 4 * (32 * vox + 8 * vox + (16 * vox2 if sib else 0))
        roof = None
        if watched:
            avg_ms = float(np.mean(watched))
            ach = flops / (avg_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': 'conv_xp_kernel<C4=4,SIB> (conv_b0_0_1: 32 warped channels -> 8, 3x3x3, full resolution, + sibling '
                                              'conv_b0_1_0: -> 16, stride 2)',
                    'achieved': round(ach, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': pmc_traffic(args),
                    'traffic_note': 'HBM bytes per launch, WRITE_SIZE + FETCH_SIZE (rocprofv3 --pmc, profiles/'
                                    'round1_pmc_dominant_kernel_v2.json); algorithmic bytes %d' % alg_bytes,
                    'avg_launch_ms': round(avg_ms, 4), 'launches': len(watched),
                    'algorithmic_flops_per_launch': flops}
        line = {
            'metric': 'depth-maps/sec at %dx%dxD=%d, N=%d views' % (args.width, args.height, args.depths, args.views),
            'value': round(args.steps / dt, 4),
            'unit': 'depth-maps/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True,
            'scaling': 'strong' if world > 1 else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '1 depth map: %d views (1 ref + %d src) %dx%d, D=%d, example.py multi-view pipeline'
                                   % (args.views, args.views - 1, args.width, args.height, args.depths),
                       'feature_hw': [h, w], 'voxels': vox,
                       'parallelism': 'views sharded over %d GPUs, RCCL all-reduce in AAM1/AAM2' % world if world > 1 else 'single GPU',
                       'launch': 'eager' if graphed is None else ('HIP graph replay, per-view streams' if world == 1 else
                                                                   'HIP graphs between the all-reduces, per-view streams')},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(args, args.cpu_baseline_seconds)
            except Exception as e:                        # the baseline must never hide the GPU number
                line['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
