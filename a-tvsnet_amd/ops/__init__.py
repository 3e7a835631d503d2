"""Python face of the C-ABI (include/atvsnet_hip.h): one function per entry point, in seven modules --

    base      ctypes plumbing, the dispatch switches (`cfg`, `configure`), launch timing
    packing   weights in operand order + their cache, tap lists, the chunk-planar layout
    geometry  homographies, warps, cost / refinement volumes, soft-argmin
    norm      batch norm, element-wise glue, the deferred batch-norm algebra (PendingBN / PendingSum / LazySlice)
    launch    dispatch policy (`*_ok`) and single-kernel launch wrappers
    convolution  `conv()` and the composite forms (SplitVolume, siblings, stems, transposed convolution)
    aanet     AANet aggregation, depth-map fusion

Every name is re-exported here (`ops.conv`, `ops.cfg`, ...): callers import this package, never a submodule.  The modules share
ONE set of state objects (`cfg`, the pack caches, the timing watch), defined once in base / packing and imported by reference.
"""
from .base import (Config, Stats, _ERR, _Timed, _call, _dev_ok, _new, _p, _ptr_array, _side_pool, _side_stream,
    _stream, _watch, _watched, cfg, configure, watch)      # noqa: F401
from .packing import (PLANAR_PAD, _Packed, _fold_cache, _pack_cache, _virt_cache, _xkind, _xp_cache, cache_snapshot,
    clear_pack_cache, conv_taps, deconv_s2_class_taps, deconv_up_ok, invalidate_weights, pack_conv3d_b, pack_conv_c16,
    pack_conv_c16b, pack_conv_weights, pack_conv_weights_tiled, pack_conv_xp, pack_conv_xp_sibling, pack_deconv_up,
    planar_cost_volume_ok, planar_pieces_decode, planar_pieces_ok, planar_stride, planar_view, same_pad, split_on)      # noqa: F401
from .geometry import (WARP_NEAREST, absdiff_mask, build_cost_volume, geo_ref_planes, geo_volume, get_homographies,
    interpolate, pixel_grids, probability_map, softargmin, tile_planes, transform_depth, transform_depth_batch,
    upsample_softargmin, visual_hull, warp_by_depth, warp_by_depth_err, warp_planes)      # noqa: F401
from .norm import (LAZY, LazySlice, PendingBN, PendingSum, _flag_pool, _param_groups, add_n, avg_pool_same, batch_norm,
    bn_add, bn_apply, bn_params, channel_stats, concat_channels, copy_channels, nonfinite_flag, nonfinite_seen,
    resize_bilinear, siblings_prologue_ok, stack)      # noqa: F401
from .launch import (Fin, XPAIR_TAPS, _fin_counter, _fin_pool, _from5, _pick_tile_m, _stats_buffer, _to5,
    _xpair_virtual_kernel, bottleneck, bottleneck_ok, conv1x1, conv1x1_ok, conv2d_lds, conv2d_lds_ok, conv2d_tail,
    conv2d_tail_ok, conv_blocks, conv_launch, conv_tiled_launch, conv_xp_launch, norm_on_load_2d_ok,
    norm_on_load_3d_ok, pack_conv1x1, pack_conv2d_lds, tiled_blocks, tiled_nsplit, tiled_tile_y, xp_blocks)      # noqa: F401
from .convolution import (SplitVolume, _DECONV_OFFSETS, _deconv_virtual_kernel, _fold_split_weights, conv, conv3d_8to1,
    conv3d_transpose_s2, conv_siblings, conv_split, conv_split_into_plane, conv_split_siblings, deconv_sum_ok,
    photo_pieces_ok, planar_concat_ok, refine_stems, siblings_ok)      # noqa: F401
from .aanet import (aanet_combine, aanet_fused, aanet_fused_ok, aanet_partial, divide, fusibile)      # noqa: F401
from .. import _lib      # noqa: F401  (ops._lib: tests and tools reach the loader through this package)
