#!/usr/bin/env python
"""Entry point mirroring the reference's ``atvsnet/example.py`` (CLI, data layout, outputs).

    python -m atvsnet_amd.atvsnet.example --root_path ../example/ --example_index 2 \
        --pretrained_model_ckpt_path weights.npz --view_num 2

* data: ``<root>/<idx>/{i.jpg, i_cam.npy, 0_gt.npy}`` (reference example.py:307-342);
* flags: the reference's (``FLAGS``: max_d, view_num, ..., example.py:25-48);
* outputs: ``result/pred.npy`` (H,W float32 depth), ``pred.jpg`` (viridis of the inverse depth),
  ``error.xlsx`` (sheet ``<view_num>_view``), reference example.py:183-213, 269-299.

What differs: there is no graph/session.  ``infer_twoview`` / ``infer_multiview`` run the whole
pipeline on one MI355X without leaving the device -- the reference bounces every per-view
(D,h,w,8) volume through host numpy between ``sess.run`` calls (example.py:144-181).  The
reference feature tower is computed once per depth map instead of once per source view.

Weights: the reference restores a TF-1.5 checkpoint that is not distributed with it.  Here
``--pretrained_model_ckpt_path`` takes an ``.npz`` of ``{tf_variable_name: array}`` (see
variables.variable_specs()); ``--synthetic_weights`` uses the seeded random initialisation.
"""
from __future__ import print_function

import argparse
import os
import sys

import numpy as np
import torch

from .. import variables
from ..flags import FLAGS
from ..tools.common import Notify
from ..tools import xlsx
from .eval_errors import acc_metrics_namelist, calc_error, err_metrics_namelist
from .model import (TVSNet, TVSNet_base_siamese, TVSNet_feature_extraction, TVSNet_refine,           # noqa: F401
                    base_stage_batch, cost_volume_aggregation, cost_volume_aggregation_refine,
                    extract_feature_shallow, feature_extraction_batch, output_conv, output_conv_refine, prob2depth,
                    prob2depth_upsample, refinement_batch, shallow_feature_batch)
from .. import ops
from ..cnn_wrapper.atvsnet import ResNetDS2SPP_shallow_f16


def depth_range(cams):
    """depth_start = cams[0,0,1,3,0], depth_interval = cams[0,0,1,3,1] as 1-element device tensors
    (reference example.py:66-69)."""
    return cams[0, 0, 1, 3, 0:1].contiguous(), cams[0, 0, 1, 3, 1:2].contiguous()


# Every network of a depth map is evaluated ONCE over all its independent calls (views, siamese directions) stacked on
# the batch axis, with per-call batch statistics (model.*_batch): ~8x fewer, 8x larger launches than the reference's
# call-per-view order, same values.  batched=False keeps the call-per-view order (per-view HIP streams).
BATCHED = True


def infer_twoview(images, cams, max_d=None, batched=None):
    """The graph of run_test_twoview (reference example.py:239-240, 267): images (1,2,H,W,3) float32
    BGR 0..255, cams (1,2,2,4,4), both on the device -> inverse-depth map (1,H,W,1) on the device."""
    max_d = FLAGS.max_d if max_d is None else max_d
    depth_start, depth_interval = depth_range(cams)
    if BATCHED if batched is None else batched:
        # model.TVSNet (reference model.py:346-377) with both towers, both siamese directions in one pass each
        feats = feature_extraction_batch(images)
        hom = {}
        _, prob_b2, depth_b2, dview = base_stage_batch(feats, cams, max_d, depth_start, depth_interval, fwd=[1], rev=[1], hom=hom)
        shallow = shallow_feature_batch(images)
        _, prob_residual = refinement_batch(depth_b2, dview, prob_b2, cams, max_d, depth_start, depth_interval, [1], shallow,
                                            hom=hom)
        refined_prob_volume = ops.add_n([prob_b2, prob_residual])
        _, depth_refined = prob2depth_upsample(refined_prob_volume, max_d, depth_start, depth_interval, out_prob_map=False)
        return depth_refined
    refined_prob_volume = TVSNet(images, cams, max_d, depth_start, depth_interval, view_i=1, ref_i=0)
    _, depth_refined = prob2depth_upsample(refined_prob_volume, max_d, depth_start, depth_interval, out_prob_map=False)
    return depth_refined


# A-B on MI355X (tools_dev/ab.py): 49.3 ms without, 52.5 ms with (five concurrent towers delay the reference
# tower every stream then waits for)
OVERLAP_REF_TOWER = False
# Upper bound on concurrently issued views (views are dealt round-robin to the streams).  A-B at config 3 (4 sources):
# 1 stream 55.3 ms, 2 streams 45.7, 3 streams 47.4, 4 streams 43.5; one stream per (source, direction) = 8 streams
# 50.1 ms -- more concurrency than one stream per source makes the GPU-filling kernels of different streams collide.
MAX_VIEW_STREAMS = 16


class _ViewStreams(object):
    """One HIP stream per source view (plus the caller's stream).

    The per-view base and refinement stages are independent (reference example.py:144-149,163-172), and
    many of their kernels (1/4- and 1/8-resolution layers, 2-D towers) cannot fill 256 CUs on their own:
    issuing the views on separate streams lets the GPU overlap them.  Every tensor a view stream produces
    is handed to the main stream with an event wait + record_stream (caching-allocator safety)."""

    def __init__(self, n, device, enabled):
        self.device = device
        self.enabled = bool(enabled) and device.type == 'cuda' and n > 1
        self.streams = [torch.cuda.Stream(device) for _ in range(min(n, MAX_VIEW_STREAMS))] if self.enabled else []

    @property
    def main(self):
        """The caller's stream NOW (a pipeline captured as several graphs re-enters with a new capture stream)."""
        return torch.cuda.current_stream(self.device) if self.device.type == 'cuda' else None

    def run(self, i, fn, after=None):
        """fn() on stream i, after everything queued so far on the main stream (or after the event `after`
        recorded earlier on it); returns fn's result."""
        if not self.enabled:
            return fn()
        st = self.streams[i % len(self.streams)]
        if after is not None:
            st.wait_event(after)
        else:
            st.wait_stream(self.main)
        with torch.cuda.stream(st):
            return fn()

    def mark(self):
        """An event on the main stream at this point of the issue order (None when streams are off)."""
        return self.main.record_event() if self.enabled else None

    def join(self, tensors):
        """Main stream waits for every view stream; `tensors` (nested lists ok) become usable on it."""
        if not self.enabled:
            return
        main = self.main
        for st in self.streams:
            main.wait_stream(st)
        if torch.cuda.is_current_stream_capturing():        # a capturing graph owns its pool's lifetimes
            return

        def rec(t):
            if isinstance(t, (list, tuple)):
                for u in t:
                    rec(u)
            elif isinstance(t, torch.Tensor):
                t.record_stream(main)
        rec(tensors)


def _infer_multiview_batched(images, cams, max_d, stages, out_prob_map):
    """infer_multiview with every per-view network evaluated once over all views (model.*_batch)."""
    n = images.shape[1]
    src = list(range(1, n))
    depth_start, depth_interval = depth_range(cams)
    feats = feature_extraction_batch(images)
    hom = {}                       # the plane sweeps of the camera pairs: computed once per depth map
    filtered, _, _, depth_view = base_stage_batch(feats, cams, max_d, depth_start, depth_interval, fwd=src, rev=src, hom=hom)
    del feats
    # AAM1
    cost_volume_agg = cost_volume_aggregation(filtered, reuse=False, keepchannel=True)
    prob_volume_agg = output_conv(cost_volume_agg, reuse=False)
    depth_agg_init = prob2depth(prob_volume_agg, max_d, depth_start, depth_interval, out_prob_map=False)
    del filtered
    # refinement of every source against the aggregated estimate
    shallow = shallow_feature_batch(images)
    # refined_cost = filtered_cost + residual (model.py:438) of every source: formed by the pass that forms the residuals
    _, _, refined = refinement_batch(depth_agg_init, depth_view, prob_volume_agg, cams, max_d, depth_start, depth_interval, src,
                                     shallow, hom=hom, residual_base=cost_volume_agg)
    # AAM2
    refined_cost_volume_agg = cost_volume_aggregation_refine(refined, reuse=False, keepchannel=True)
    refined_prob_volume_agg = output_conv_refine(refined_cost_volume_agg, reuse=False)
    final = prob2depth_upsample(refined_prob_volume_agg, max_d, depth_start, depth_interval, out_prob_map=out_prob_map)
    if stages is not None:
        stages.update(depth_views=[depth_view[v] for v in src], cost_volume_agg=cost_volume_agg,
                      prob_volume_agg=prob_volume_agg, depth_agg_init=depth_agg_init,
                      refined_cost_volume_agg=refined_cost_volume_agg, refined_prob_volume_agg=refined_prob_volume_agg)
    return final if out_prob_map else final[1]


def infer_multiview(images, cams, max_d=None, stages=None, view_streams=True, out_prob_map=False, batched=None):
    """The run loop of run_test_multiview (reference example.py:140-181), on the device:
    base (per source) -> AAM1 -> refinement (per source) -> AAM2 -> x4 upsample + soft-argmin.
    batched (default BATCHED): one pass of each network over all its per-view calls; otherwise call per view, and
    view_streams: issue the independent per-view stages on separate HIP streams.
    out_prob_map: return (depth, depth_up, prob_map, prob_map_up) as the ETH3D driver's last stage does
    (reference eval_pointcloud.py:268-272) instead of depth_up alone."""
    max_d = FLAGS.max_d if max_d is None else max_d
    n = images.shape[1]
    assert n > 2
    if BATCHED if batched is None else batched:
        return _infer_multiview_batched(images, cams, max_d, stages, out_prob_map)
    depth_start, depth_interval = depth_range(cams)
    vs = _ViewStreams(n - 1, images.device, view_streams)
    start = vs.mark() if OVERLAP_REF_TOWER else None     # the source towers need not wait for the reference tower ...
    ref_feature = TVSNet_feature_extraction(images, 0)
    ref_ready = vs.mark() if OVERLAP_REF_TOWER else None  # ... only their cost volumes do
    base = [vs.run(v - 1, lambda v=v: TVSNet_base_siamese(images, cams, max_d, depth_start, depth_interval, view_i=v,
                                                          ref_i=0, ref_feature=ref_feature, ref_ready=ref_ready),
                   after=start)
            for v in range(1, n)]
    vs.join(base)
    filtered_cost_volumes = [b[2] for b in base]    # prob volumes are fed but unused by the reference (quirk C12)
    depth_views = [b[3] for b in base]
    del base
    # AAM1
    cost_volume_agg = cost_volume_aggregation(filtered_cost_volumes, reuse=False, keepchannel=True)
    prob_volume_agg = output_conv(cost_volume_agg, reuse=False)
    depth_agg_init = prob2depth(prob_volume_agg, max_d, depth_start, depth_interval, out_prob_map=False)
    del filtered_cost_volumes
    # refinement against the aggregated estimate
    ref_shallow = ResNetDS2SPP_shallow_f16({'data': images[:, 0]}, is_training=True).get_output()

    def refine(view_i):
        shallow = extract_feature_shallow(images, 0, view_i, ref_feature=ref_shallow)
        return TVSNet_refine(depth_agg_init, depth_views[view_i - 1], prob_volume_agg, cost_volume_agg, images, cams,
                             max_d, depth_start, depth_interval, view_i=view_i, ref_i=0, shallow_features=shallow)[1]
    refined_cost_volumes = [vs.run(v - 1, lambda v=v: refine(v)) for v in range(1, n)]
    vs.join(refined_cost_volumes)
    # AAM2
    refined_cost_volume_agg = cost_volume_aggregation_refine(refined_cost_volumes, reuse=False, keepchannel=True)
    refined_prob_volume_agg = output_conv_refine(refined_cost_volume_agg, reuse=False)
    final = prob2depth_upsample(refined_prob_volume_agg, max_d, depth_start, depth_interval, out_prob_map=out_prob_map)
    depth_agg_refined = final[1]
    if stages is not None:
        stages.update(depth_views=depth_views, cost_volume_agg=cost_volume_agg, prob_volume_agg=prob_volume_agg,
                      depth_agg_init=depth_agg_init, refined_cost_volume_agg=refined_cost_volume_agg,
                      refined_prob_volume_agg=refined_prob_volume_agg)
    return final if out_prob_map else depth_agg_refined


class GraphedInference(object):
    """The whole depth-map pipeline captured once in a HIP graph and replayed per depth map.

    Eager execution issues ~2000 kernel launches per depth map from Python (~20 us each), which is
    close to the GPU time of the step; a captured graph (all shapes are static for a given
    (views, H, W, D)) replays them -- including the per-view stream fork/join -- with one host call.
    Inputs live in static device buffers: pass new images / cams to __call__ to overwrite them.
    """

    def __init__(self, images, cams, max_d=None, view_streams=True, out_prob_map=False, batched=None):
        from .. import ops
        self.max_d = FLAGS.max_d if max_d is None else max_d
        self.split16 = bool(ops.cfg.split16)    # the kernels this graph was captured with (a replay ignores later switches)
        self._fp32 = None                       # the same pipeline captured on the fp32 matrix cores, built on first need
        self.out_prob_map = out_prob_map
        self.batched = BATCHED if batched is None else batched
        self.images = images.clone()
        self.cams = cams.clone()
        self.twoview = images.shape[1] == 2
        self.view_streams = view_streams
        side = torch.cuda.Stream(images.device)
        side.wait_stream(torch.cuda.current_stream(images.device))
        with torch.cuda.stream(side):          # warm-up: weight packing / uploads, function attributes
            self._run()
        torch.cuda.current_stream(images.device).wait_stream(side)
        torch.cuda.synchronize(images.device)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog of a multi-GPU run) may issue HIP calls meanwhile
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.out = self._run()
        # the graph holds raw pointers to the weights it was captured with: keep those copies alive even if the
        # variable store is reloaded afterwards (the graph then goes on computing with the captured weights)
        self._weights = (ops.cache_snapshot(), variables.default_store().device_snapshot())

    def _run(self):
        if self.twoview:
            return infer_twoview(self.images, self.cams, self.max_d, batched=self.batched)
        return infer_multiview(self.images, self.cams, self.max_d, view_streams=self.view_streams,
                               out_prob_map=self.out_prob_map, batched=self.batched)

    def __call__(self, images=None, cams=None):
        if images is not None:
            self.images.copy_(images)
        if cams is not None:
            self.cams.copy_(cams)
        self.graph.replay()
        return self.out

    def checked(self, images=None, cams=None):
        """__call__, then wait for the depth map; if a batch norm of this replay saw non-finite moments (an fp16-range overflow
        of the split-operand kernels) THAT map is recomputed on the fp32 matrix cores (`fp32_rerun`) and the fp32 result is
        returned -- the reference is fp32 end to end (cnn_wrapper/network.py:165-167, 570-601), a drop-in must not need a
        user action to have fp32's range.  FloatingPointError only if the fp32 kernels see non-finite values too."""
        out = self(images, cams)
        dev = self.images.device
        from .. import ops
        if ops.nonfinite_seen(dev):
            out = self.fp32_rerun()
        return out

    def fp32_rerun(self):
        """The depth map of the inputs now in this graph's static buffers on the fp32-MFMA kernels: a second captured graph
        (built once, in this process -- never a re-exec of a process that has touched the GPU), replayed synchronously.
        Raises FloatingPointError if this graph already IS the fp32 form or the fp32 kernels see non-finite moments too."""
        from .. import ops
        dev = self.images.device
        if not self.split16:
            raise FloatingPointError('a batch norm saw non-finite moments on the fp32 kernels: ' + _NONFINITE_HINT)
        _log_fp32_fallback()
        if self._fp32 is None:
            with ops.configure(split16=False):
                self._fp32 = GraphedInference(self.images, self.cams, self.max_d, view_streams=self.view_streams,
                                              out_prob_map=self.out_prob_map, batched=self.batched)
            ops.nonfinite_seen(dev)            # the capture's warm-up ran on the inputs too: start from a clear flag
        out = self._fp32(self.images, self.cams)
        if ops.nonfinite_seen(dev):
            raise FloatingPointError('a batch norm saw non-finite moments on the fp32 kernels too: ' + _NONFINITE_HINT)
        return out


_hip_runtime = None


def cu_split_streams(device, parts):
    """`parts` HIP streams whose kernels run on DISJOINT sets of compute units: part k of every XCD (hipExtStreamCreateWithCUMask;
    mask bit i = CU i / 8 of XCD i % 8 on MI355X, so part k owns the CUs with (i / 8) % parts == k -- an equal share of every
    XCD, of its L2 and of the memory channels behind it).  Kernels of different streams then never share a CU, hence never a
    SIMD: the one condition of the co-residency fault cannot arise between them.
    (Masks that leave an XCD without a CU -- "even / odd bits" -- are not honoured by the runtime: the stream then runs on the whole
    chip.  tests/test_gpu_pipeline.py checks that a stream of this function really is confined.)  The streams live as long as the
    process (a handful per process; nothing to free in a driver run)."""
    import ctypes
    global _hip_runtime
    props = torch.cuda.get_device_properties(device)
    ncu, nxcd = int(props.multi_processor_count), 8
    if parts < 1 or ncu % nxcd or (ncu // nxcd) < parts:
        raise ValueError('cu_split_streams: %d parts of %d compute units' % (parts, ncu))
    if _hip_runtime is None:
        # the HIP runtime torch has ALREADY loaded (a stream of a second copy of the runtime would mean nothing to torch): its path from
        # this process's mappings, the soname as a fallback (dlopen returns the loaded instance for a matching soname)
        path = 'libamdhip64.so'
        try:
            with open('/proc/self/maps') as f:
                for ln in f:
                    if 'libamdhip64.so' in ln:
                        path = ln.split()[-1]
                        break
        except OSError:
            pass
        _hip_runtime = ctypes.CDLL(path)
    out = []
    with torch.cuda.device(device):
        for k in range(parts):
            words = (ctypes.c_uint32 * ((ncu + 31) // 32))()
            for i in range(ncu):
                if (i // nxcd) % parts == k:
                    words[i // 32] |= (1 << (i % 32))
            st = ctypes.c_void_p()
            rc = _hip_runtime.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), words)
            if rc != 0 or not st.value:
                raise RuntimeError('hipExtStreamCreateWithCUMask failed (%d)' % rc)
            out.append(torch.cuda.ExternalStream(st.value, device=device))
    return out


class PipelinedInference(object):
    """`slots` depth maps queued: one captured graph (static buffers) and one HIP stream per slot.

    The depth maps of a scene are independent (one per reference view, reference eval_pointcloud.py:399-424).
    `co_resident=False` (default): a slot's graph starts when the previously submitted one has finished -- the GPU runs
    ONE depth map at a time, bit for bit the single-map path, and what the queue buys is that the host prepares and submits
    the next map (and writes the previous one's files) meanwhile.
    `co_resident=True`: the slots' graphs run concurrently on their streams; the second map's kernels fill the phases in
    which one pipeline leaves the GPU under-filled (+4.5 % depth maps/s at config 3).  Opt-in only: wavefronts of different
    kernels then share SIMDs, and on this pool's MI355X kernels with compiler-formed packed fp32 arithmetic on dwordx2-loaded
    operands have produced wrong lane quarters beside another kernel's 16x16x32 MFMA wavefronts (narrowed
    to that instruction form, cause not established; the kernels that still contain packed fp32 are pinned by
    tests/test_packed_fp32_census.py, and conv2d_b / conv1x1_b / bottleneck_b / deconv_up_b run two workgroups per CU, so
    nothing reserves their SIMDs).  bench.py measures it under `pipelined` and fails the run if a slot's output differs from
    the single-map output.
    `co_resident='cu_split'` (round 6): the slots' graphs run concurrently, each on its OWN share of every XCD's compute units
    (cu_split_streams): no SIMD ever holds wavefronts of two kernels, so the fault above cannot occur, and every slot still
    produces the single-map bits.  Two slots: +1.5 ... 4 % depth maps/s at config 3 (each map has half the chip; what is gained
    is the overlap of one map's launch tails and latency-bound kernels with the other's work), +19 % for two-view maps
    (configs[1]: 184 -> 218 maps/s); the latency of ONE map roughly doubles.  bench.py reports it under `pipelined_cu_split`.

        t = p.submit(images, cams)      # asynchronous: copies the inputs, replays the slot's graph on its stream
        out = p.result(t)               # waits for that depth map; the tensors are valid until the slot is re-used
    """

    def __init__(self, images, cams, max_d=None, slots=2, co_resident=False, **kw):
        if slots < 1:
            raise ValueError('PipelinedInference: slots >= 1')
        self.device = images.device
        if co_resident not in (False, True, 'cu_split'):
            raise ValueError("PipelinedInference: co_resident is False, True or 'cu_split'")
        self.cu_split = co_resident == 'cu_split'
        self.co_resident = bool(co_resident)
        self.last = None                 # slot of the most recent submission (its event orders the next one behind it)
        self.graphs = [GraphedInference(images, cams, max_d, **kw) for _ in range(slots)]
        self.streams = cu_split_streams(self.device, slots) if self.cu_split else [torch.cuda.Stream(self.device) for _ in range(slots)]
        self.events = [torch.cuda.Event() for _ in range(slots)]
        self.busy = [False] * slots
        self.suspect = set()             # slots in flight when the non-finite flag was found set: their maps are recomputed in fp32
        self.next = 0

    @property
    def slots(self):
        return len(self.graphs)

    def set_mode(self, co_resident):
        """Switch the way the slots share the GPU (False | True | 'cu_split') with nothing in flight: the captured graphs stay,
        the slots' streams are replaced."""
        if any(self.busy):
            raise RuntimeError('PipelinedInference.set_mode: results still in flight')
        if co_resident not in (False, True, 'cu_split'):
            raise ValueError("PipelinedInference: co_resident is False, True or 'cu_split'")
        torch.cuda.synchronize(self.device)
        split = co_resident == 'cu_split'
        if split != self.cu_split:
            self.streams = cu_split_streams(self.device, self.slots) if split else [torch.cuda.Stream(self.device) for _ in range(self.slots)]
        self.cu_split, self.co_resident, self.last = split, bool(co_resident), None

    def submit(self, images=None, cams=None):
        """Issue one depth map on the next slot (its previous result must have been fetched); returns the ticket."""
        s = self.next
        if self.busy[s]:
            raise RuntimeError('PipelinedInference: slot %d still holds an unfetched result' % s)
        self.next = (s + 1) % len(self.graphs)
        st = self.streams[s]
        # inputs prepared on the caller's stream are ordered in front of the slot's work.  cu_split: the CU-masked streams are BLOCKING
        # streams in the legacy sense -- any operation on the default stream (an event record, a copy) waits for every map in flight
        # and holds the next one back, which serialises the slots (34 instead of 62 maps/s at configs[2]); so HOST tensors (or None)
        # are copied by the slot's own stream with no default-stream operation at all, and only device inputs pay for the ordering
        if not self.cu_split or any(t is not None and t.is_cuda for t in (images, cams)):
            cur = torch.cuda.current_stream(self.device)
            if self.cu_split and cur == torch.cuda.default_stream(self.device) and not getattr(self, '_warned', False):
                self._warned = True
                print(Notify.WARNING, "PipelinedInference(co_resident='cu_split'): device inputs prepared on the default stream order "
                      'every submission behind ALL maps in flight (the slots then run one after the other, each on its share of the '
                      'chip); pass host tensors or prepare the inputs on a side stream', Notify.ENDC)
            st.wait_stream(cur)
        if not self.co_resident and self.last is not None and self.last != s:
            st.wait_event(self.events[self.last])                   # one depth map on the GPU at a time
        for t in (images, cams):
            # the copy into the slot's static buffers runs on the slot's stream, possibly long after this call returns:
            # tell the caching allocator, or the caller's next allocation could re-use the block while it is still read
            if t is not None and t.is_cuda:
                t.record_stream(st)
        with torch.cuda.stream(st):
            self.graphs[s](images, cams)
            self.events[s].record(st)
        self.busy[s] = True
        self.last = s
        return s

    def result(self, ticket, host=False):
        """The depth map (tuple of outputs with out_prob_map) of `ticket`; host=True: as CPU tensors, copied by the slot's own stream
        (cu_split: the way to fetch results without an operation on the default stream, see submit)."""
        if not self.busy[ticket]:
            raise RuntimeError('PipelinedInference: nothing in flight on slot %d' % ticket)
        self.events[ticket].synchronize()
        with torch.cuda.stream(self.streams[ticket]):          # the flag read and the copies below: on the slot's (idle) stream
            out = self._result(ticket)
            if host:
                out = tuple(o.cpu() for o in out) if isinstance(out, (tuple, list)) else out.cpu()
        return out

    def _result(self, ticket):
        self.busy[ticket] = False
        from .. import ops
        if ops.nonfinite_seen(self.device):
            # an fp16-range overflow of the split-operand kernels is never returned.  The sticky flag does not say WHICH of the
            # maps in flight set it: every map that was in flight when it is found up is recomputed on the fp32 kernels
            # (GraphedInference.fp32_rerun: synchronous, from the slot's static input buffers)
            torch.cuda.synchronize(self.device)
            ops.nonfinite_seen(self.device)
            self.suspect.update(t for t, b in enumerate(self.busy) if b)
            self.suspect.add(ticket)
        if ticket in self.suspect:
            self.suspect.discard(ticket)
            return self.graphs[ticket].fp32_rerun()
        return self.graphs[ticket].out

    def run(self, count):
        """Benchmark helper: `count` depth maps of the captured inputs, round-robin over the slots; returns when all are
        done (results are overwritten)."""
        for s, st in enumerate(self.streams):
            st.wait_stream(torch.cuda.current_stream(self.device))
        for i in range(count):
            s = i % len(self.graphs)
            if not self.co_resident and self.last is not None and self.last != s:
                self.streams[s].wait_event(self.events[self.last])
            with torch.cuda.stream(self.streams[s]):
                self.graphs[s].graph.replay()
                self.events[s].record(self.streams[s])
            self.last = s
        # wait on the HOST for every slot's last event -- not `current_stream.wait_stream(slot stream)`: the CU-masked streams are
        # blocking streams in the legacy sense, and an operation on the default stream while their queues are full cost 13 % of the
        # run's throughput (round 6, tools_dev/cu_mask_probe.py: 61.4 -> 53.9 maps/s; plain side streams are unaffected)
        for s in range(min(count, len(self.graphs))):
            self.events[s].synchronize()


def _load_weights():
    path = FLAGS.pretrained_model_ckpt_path
    store = variables.default_store()
    if getattr(FLAGS, 'synthetic_weights', False):
        store.init_synthetic(1234)
        print(Notify.INFO, 'using seeded synthetic weights (no checkpoint)', Notify.ENDC)
        return
    if path is None:
        print('FLAGS.pretrained_model_ckpt_path is None !!')
        sys.exit()
    if path.endswith('.npz'):
        store.load_npz(path)             # {tf_variable_name: array}
    elif os.path.exists(path + '.index'):
        store.load_checkpoint(path)      # the TensorFlow checkpoint the reference restores (tools/tf_checkpoint.py)
    else:
        raise RuntimeError('%s: neither an .npz of {tf_variable_name: array} nor a TensorFlow checkpoint prefix '
                           '(%s.index not found); pass --synthetic_weights to run without weights' % (path, path))
    print(Notify.INFO, 'pre-trained model restored from %s' % path, Notify.ENDC)


def _save_results(savepath, out_depth_map, out_disp_map, depth_gt):
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    if not os.path.exists(savepath):
        os.makedirs(savepath)
    np.save(os.path.join(savepath, 'pred.npy'), np.squeeze(np.array(out_depth_map)))
    plt.imsave(os.path.join(savepath, 'pred.jpg'), np.squeeze(np.array(out_disp_map)), cmap='viridis')
    if depth_gt is not None:
        print(Notify.INFO, 'calulating error......', Notify.ENDC)
        error, _ = calc_error(np.squeeze(out_depth_map), np.squeeze(depth_gt))
        write_error_xlsx(os.path.join(savepath, 'error.xlsx'), error, FLAGS.view_num)
    print(Notify.INFO, "result save to {}.".format(savepath), Notify.ENDC)


def write_error_xlsx(path, error, view_num):
    """The sheet layout of reference example.py:199-213."""
    workbook = xlsx.Workbook(path)
    worksheet = workbook.add_worksheet(str(view_num) + '_view')
    n_err = len(err_metrics_namelist)
    for i, name in enumerate(err_metrics_namelist):
        worksheet.write(i + 1, 0, name)
    for i, name in enumerate(acc_metrics_namelist):
        worksheet.write(i + n_err + 2, 0, name)
    values = error.tolist()
    worksheet.write(0, 1, 'err')
    worksheet.write(n_err + 1, 1, 'acc')
    for i in range(n_err):
        worksheet.write(i + 1, 1, values[i])
    for i in range(n_err, len(values)):
        worksheet.write(i + 2, 1, values[i])
    workbook.close()


_NONFINITE_HINT = 'the inputs or the weights are not finite (the fp32 kernels have the reference\'s range)'
_fallback_logged = [False]


def _log_fp32_fallback():
    if not _fallback_logged[0]:
        _fallback_logged[0] = True
        print(Notify.INFO, 'an activation left the fp16 range of the split-operand kernels: this depth map is recomputed on the '
              'fp32 matrix cores (ATVS_SPLIT16=0 selects them from the start)', Notify.ENDC)


def infer_checked(fn, device=None):
    """fn() -> device tensor(s), with the range guard of the host drivers: if a batch norm saw non-finite moments (the sticky
    device flag) or an output is not finite, fn() runs again under ops.configure(split16=False) -- every convolution on the
    fp32 matrix cores, the reference's arithmetic range (cnn_wrapper/network.py:165-167) -- and THAT result is returned.
    FloatingPointError only if the fp32 kernels fail as well (non-finite inputs / weights).  Synchronises."""
    from .. import ops
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else device

    def finite(out):
        ts = out if isinstance(out, (list, tuple)) else [out]
        return all(bool(torch.isfinite(t).all()) for t in ts if isinstance(t, torch.Tensor))
    ops.nonfinite_seen(device)                 # start from a clear flag: only THIS map's batch norms count
    out = fn()
    if not ops.nonfinite_seen(device) and finite(out):
        return out
    if not ops.cfg.split16:
        raise FloatingPointError('non-finite values on the fp32 kernels: ' + _NONFINITE_HINT)
    _log_fp32_fallback()
    del out
    with ops.configure(split16=False):
        out = fn()
        bad = ops.nonfinite_seen(device) or not finite(out)
    if bad:
        raise FloatingPointError('non-finite values on the fp32 kernels too: ' + _NONFINITE_HINT)
    return out


_RANGE_HINT = ('an activation or weight left the fp16 range of the split-operand kernels (or the inputs were not finite); rerun with '
               'ATVS_SPLIT16=0 for the fp32 kernels')


def check_device(device=None):
    """Raise if a batch norm on `device` saw a non-finite moment since the last check (the sticky flag atvs_bn_finalize sets,
    ops.nonfinite_seen): catches an fp16-range overflow of the split-operand kernels even where a later ReLU swallowed the NaN
    before it could reach the depth map.  Synchronises with the device; GraphedInference.checked() / PipelinedInference.result()
    and the host drivers call it on every result they hand out."""
    from .. import ops
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
    if ops.nonfinite_seen(device):
        raise FloatingPointError('a batch norm saw non-finite moments: ' + _RANGE_HINT)


def check_finite(arr, what='depth map'):
    """The split-operand convolutions carry activations as two fp16 pieces: a value beyond +-65504 turns
    into inf/NaN there instead of a silently wrong depth.  The host drivers call this on every result they copy back so that
    the failure names its cause (ATVS_SPLIT16=0 selects the fp32 matrix-core kernels, which have fp32's range)."""
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        check_device()
    if not np.isfinite(arr).all():
        raise FloatingPointError('%s holds %d non-finite values: an activation or weight left the fp16 range of the split-operand '
                                 'kernels (or the inputs were not finite); rerun with ATVS_SPLIT16=0 for the fp32 kernels'
                                 % (what, int((~np.isfinite(arr)).sum())))
    return arr


def _to_device(images_data, cams_data):
    torch.cuda.set_device(FLAGS.gpu_id)          # every kernel launches on the current device's stream
    dev = torch.device('cuda:%d' % FLAGS.gpu_id)
    images = torch.from_numpy(np.ascontiguousarray(images_data, dtype=np.float32))[None].to(dev)
    cams = torch.from_numpy(np.ascontiguousarray(cams_data, dtype=np.float32))[None].to(dev)
    return images, cams


def run_test_multiview(savepath, images_data, cams_data, depth_gt=None):
    """reference example.py:51-216."""
    assert FLAGS.view_num > 2
    print(Notify.INFO, 'loading checkpoint......', Notify.ENDC)
    _load_weights()
    images, cams = _to_device(images_data, cams_data)
    print(Notify.INFO, 'running test......', Notify.ENDC)
    out_depth_map = check_finite(infer_checked(lambda: infer_multiview(images, cams, FLAGS.max_d)).cpu().numpy())
    out_disp_map = out_depth_map.copy()
    if FLAGS.inverse_depth:
        out_depth_map[out_depth_map < 1e-10] = float("inf")
        out_depth_map = 1.0 / out_depth_map
    _save_results(savepath, out_depth_map, out_disp_map, depth_gt)


def run_test_twoview(savepath, images_data, cams_data, depth_gt=None):
    """reference example.py:219-302."""
    assert FLAGS.view_num == 2
    print(Notify.INFO, 'loading checkpoint......', Notify.ENDC)
    _load_weights()
    images, cams = _to_device(images_data, cams_data)
    print(Notify.INFO, 'running test......', Notify.ENDC)
    out_depth_map = check_finite(infer_checked(lambda: infer_twoview(images, cams, FLAGS.max_d)).cpu().numpy())
    out_disp_map = out_depth_map.copy()
    if FLAGS.inverse_depth:
        out_depth_map[out_depth_map <= 0] = float("inf")
        out_depth_map = 1.0 / out_depth_map
    _save_results(savepath, out_depth_map, out_disp_map, depth_gt)


def _imread_bgr(path):
    """cv2.imread equivalent (BGR uint8); the image has no OpenCV, PIL decodes the JPEG."""
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1].copy()


def load_example(data_root, view_num):
    """-> (images (N,H,W,3) uint8 BGR, cams (N,2,4,4), depth_gt or None, views found) (reference :312-342)."""
    valid = 0
    for view_i in range(view_num):
        img_path = os.path.join(data_root, str(view_i) + '.jpg')
        cam_path = os.path.join(data_root, str(view_i) + '_cam.npy')
        if os.path.exists(img_path) and os.path.exists(cam_path):
            valid += 1
        else:
            print("{} or {} not exist. check view_num".format(img_path, cam_path))
    images = np.stack([_imread_bgr(os.path.join(data_root, '%d.jpg' % i)) for i in range(valid)], axis=0)
    cams = np.stack([np.load(os.path.join(data_root, '%d_cam.npy' % i)) for i in range(valid)], axis=0)
    gt_path = os.path.join(data_root, '0_gt.npy')
    depth_gt = np.load(gt_path) if os.path.exists(gt_path) else None
    return images, cams, depth_gt, valid


def main(argv=None):
    data_root = os.path.join(FLAGS.root_path, str(FLAGS.example_index))
    savepath = os.path.join(data_root, 'result')
    if not os.path.exists(savepath):
        os.makedirs(savepath)
    images, cams, depth_gt, valid = load_example(data_root, FLAGS.view_num)
    if valid != FLAGS.view_num:
        print(Notify.INFO, 'only %d views found (FLAGS.view_num = %d), continue with %d views' %
              (valid, FLAGS.view_num, valid), Notify.ENDC)
        FLAGS.view_num = valid
    if FLAGS.view_num == 2:
        run_test_twoview(savepath, images, cams, depth_gt)
    else:
        run_test_multiview(savepath, images, cams, depth_gt)


def cli(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--root_path', type=str, default=FLAGS.root_path)
    parser.add_argument('--example_index', type=int, default=FLAGS.example_index)
    parser.add_argument('--pretrained_model_ckpt_path', type=str, default=FLAGS.pretrained_model_ckpt_path)
    parser.add_argument('--view_num', type=int, default=FLAGS.view_num)
    parser.add_argument('--max_d', type=int, default=FLAGS.max_d)
    parser.add_argument('--gpu_id', type=int, default=FLAGS.gpu_id)
    parser.add_argument('--synthetic_weights', action='store_true')
    args = parser.parse_args(argv)
    for k, v in vars(args).items():
        setattr(FLAGS, k, v)
    assert FLAGS.view_num > 1
    print(Notify.INFO, 'Testing A-TVSNet with %d views' % (FLAGS.view_num), Notify.ENDC)
    main()


if __name__ == '__main__':
    cli()
