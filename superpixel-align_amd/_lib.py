"""ctypes binding of libspalign.so (the C ABI declared in include/spalign.h).

The library is the only implementation of the hot path: when it is missing or no gfx950
device is visible every op raises — there is no CPU fallback behind these functions.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SPA_LIB_PATH') or os.path.join(_HERE, 'libspalign.so')      # SPA_LIB_PATH: A/B runs of two builds on one box
_LIB = None

SPA_OK = 0
STATUS_BITS = {
    0x01: 'SLIC: a seed lost all of its pixels (NaN centre, dead for the rest of the sweeps, as in scikit-image)',
    0x02: 'SLIC: a pixel fell outside every 2S search window',
    0x04: 'connectivity: a component reached max_size (cut in BFS order by the exact replay path)',
    0x08: 'mean pooling: more than 16 superpixels touch one feature pixel',
    0x10: 'k-means: grid barrier timed out',
    0x20: 'a superpixel label outside [0, n_labels) was met',
    0x40: 'anchor selection: the device random stream ran dry',
    0x80: 'k-means initial assignment on the device: more than 65 536 points at or below the median weight',
}

# informational bits: the condition is handled exactly like the reference handles it; never an error
INFO_BITS = 0x01 | 0x04

c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_f64 = ctypes.c_double
c_p = ctypes.c_void_p


class SlicPlan(ctypes.Structure):
    _fields_ = [('n_centroids', c_i32), ('grid_ny', c_i32), ('grid_nx', c_i32),
                ('start_y', c_i32), ('start_x', c_i32), ('step_y', c_i32), ('step_x', c_i32),
                ('win_step_y', c_i32), ('win_step_x', c_i32), ('step', c_f32),
                ('min_size', c_i32), ('max_size', c_i32), ('max_labels', c_i32)]


class FmapDesc(ctypes.Structure):
    _fields_ = [('C', c_i32), ('fh', c_i32), ('fw', c_i32),
                ('stride_b', c_i64), ('stride_c', c_i64), ('stride_y', c_i64), ('stride_x', c_i64),
                ('dtype', c_i32)]


# name -> (restype, argtypes); also the list of symbols the header declares
PROTOTYPES = {
    'spa_version': (ctypes.c_int, []),
    'spa_last_error': (ctypes.c_char_p, []),
    'spa_ctx_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_p)]),
    'spa_ctx_destroy': (None, [c_p]),
    'spa_status': (ctypes.c_int, [c_p, ctypes.POINTER(ctypes.c_uint32), c_p]),
    'spa_status_peek_async': (ctypes.c_int, [c_p, c_p, c_p]),
    'spa_status_take_async': (ctypes.c_int, [c_p, c_p, c_p]),
    'spa_prof_enable': (ctypes.c_int, [c_p, ctypes.c_int]),
    'spa_prof_slots': (ctypes.c_int, []),
    'spa_prof_name': (ctypes.c_char_p, [ctypes.c_int]),
    'spa_prof_read': (ctypes.c_int, [c_p, ctypes.c_int, ctypes.POINTER(c_f64),
                                     ctypes.POINTER(ctypes.c_int)]),
    'spa_drn_normalise': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_i32, c_p, c_p, c_p]),
    'spa_bias_act': (ctypes.c_int, [c_p, c_p, c_i32, c_i64, c_i32, c_p, c_p, c_i32, c_p]),
    'spa_drn_stem_d': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_p, c_p]),
    'spa_conv3x3_bf16': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p]),
    'spa_conv_bf16_light': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p]),
    'spa_conv3x3_f32': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p]),
    'spa_conv1x1_f32': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p, c_p, c_i32, c_p, c_p]),
    'spa_wino_tiles': (ctypes.c_int64, [c_i32, c_i32, c_i32, c_i32]),
    'spa_conv3x3_wino_f32': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'spa_wino4_tiles': (ctypes.c_int64, [c_i32, c_i32, c_i32, c_i32]),
    'spa_conv3x3_wino4_f32': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'spa_conv3x3_f16s': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_f32, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'spa_conv1x1_f16s': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_f32, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p, c_p]),
    'spa_conv3x3_s2_f16s': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_f32, c_i32, c_i32, c_p, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'spa_bias_act_amax': (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_p, c_p, c_i32, c_p, c_p]),
    'spa_drn_stem_d_amax': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'spa_drn_stem_c_amax': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'spa_drn_stem_c_bf16': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'spa_conv_small_f16s': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_f32, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'spa_drn_layer2_f16s': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_f32, c_p, c_p, c_p, c_p, c_p]),
    'spa_drn_layer2_f32': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'spa_conv3x3_s2_f32': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_i32, c_p, c_p, c_p]),
    'spa_amax_f32': (ctypes.c_int, [c_p, c_p, c_i64, c_p, c_p]),
    'spa_conv3x3_wino4_f16s': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'spa_resize_bicubic_u8': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p]),
    'spa_resize_cvcubic_u8': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p]),
    'spa_debug_peek': (ctypes.c_int, [c_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, c_p]),
    'spa_debug_set': (ctypes.c_int, [c_p, ctypes.c_int32, ctypes.c_int32]),
    'spa_ws_generation': (ctypes.c_int, [c_p]),
    'spa_debug_lds_probe': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p]),
    'spa_slic_make_plan':(ctypes.c_int, [c_i32, c_i32, c_i32, ctypes.POINTER(SlicPlan)]),
    'spa_rgb2lab': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_f32, c_p, c_p]),
    'spa_slic_core': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'spa_enforce_connectivity': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p,
                                                c_p, c_p]),
    'spa_slic': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_p, c_p, c_p]),
    'spa_felzenszwalb': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_f64, c_f64, c_i32, c_p, c_p, c_p]),
    'spa_rgb2lab_u8_f64': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_f64, c_p, c_p]),
    'spa_slic_core_f64': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'spa_slic_u8': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f64, c_i32, c_p, c_p, c_p]),
    'spa_felzenszwalb_u8': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_f64, c_f64, c_i32, c_p, c_p, c_p]),
    'spa_overlap_refine': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i64, c_i32, c_f64, c_p, c_p]),
    'spa_segment_offsets':(ctypes.c_int, [c_p, c_p, c_i32, c_p, c_p]),
    'spa_segment_stats': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_i32, c_f64, c_f64,
                                         c_f64, c_f64, c_p, c_p, c_p, c_p]),
    'spa_pyrandom_create': (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(c_p)]),
    'spa_pyrandom_destroy': (None, [c_p]),
    'spa_pyrandom_shuffle_select_host': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_p, c_p]),
    'spa_nprandom_create': (ctypes.c_int, [ctypes.c_uint32, ctypes.POINTER(c_p)]),
    'spa_nprandom_destroy': (None, [c_p]),
    'spa_nprandom_shuffle_host': (ctypes.c_int, [c_p, c_p, c_i64]),
    'spa_nprandom_state': (ctypes.c_int, [c_p, c_p]),
    'spa_nprandom_set_state': (ctypes.c_int, [c_p, c_p]),
    'spa_np_kmeans_init_dev': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'spa_kmeans_retry_update': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'spa_pyrandom_dev_seed': (ctypes.c_int, [c_p, ctypes.c_uint64, c_p]),
    'spa_pyrandom_dev_generate': (ctypes.c_int, [c_p, c_i64, c_p]),
    'spa_anchor_ranks_dev': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i64, c_p, c_p, c_p]),
    'spa_select_anchor_pixels': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_i32, c_p, c_p,
                                                c_i32, c_p, c_p]),
    'spa_pool_anchor': (ctypes.c_int, [c_p, c_p, ctypes.POINTER(FmapDesc), c_i32, c_i32, c_p, c_i32,
                                       c_p, c_p, c_i32, c_i32, c_p, c_i32, c_p, c_i32, c_i64, c_p]),
    'spa_pool_mean': (ctypes.c_int, [c_p, c_p, ctypes.POINTER(FmapDesc), c_p, c_i32, c_i32, c_i32,
                                     c_p, c_i32, c_p, c_i32, c_p, c_i32, c_p, c_i32, c_i64, c_p]),
    'spa_kmeans_weighted': (ctypes.c_int, [c_p, c_p, c_i32, c_i64, c_i32, c_p, c_p, c_i32, c_i32,
                                           c_i32, c_p, c_p, c_p, c_p]),
    'spa_kmeans_weighted_gated': (ctypes.c_int, [c_p, c_p, c_i32, c_i64, c_i32, c_p, c_p, c_i32, c_i32,
                                                 c_i32, c_p, c_p, c_p, c_p, c_p]),
    'spa_paint': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'spa_confusion': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i64, c_p, c_p]),
}


class SpalignError(RuntimeError):
    pass


def build(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into libspalign.so (in tree)."""
    csrc = os.path.join(_HERE, 'csrc')
    if force:
        subprocess.check_call(['make', '-s', '-C', csrc, 'clean'])
    subprocess.check_call(['make', '-s', '-j8', '-C', csrc])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SpalignError('%s is missing: run `python -c "import __graft_entry__ as g; '
                               'g.build()"` (there is no CPU fallback)' % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64; libspalign receives torch's device
        # pointers and streams, so both must live in ONE HIP runtime: import torch first, then
        # libspalign's NEEDED libamdhip64.so.7 resolves to the runtime torch already loaded.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != SPA_OK:
        raise SpalignError('libspalign error %d: %s' % (rc, lib().spa_last_error().decode()))


def make_plan(H, W, n_segments):
    p = SlicPlan()
    check(lib().spa_slic_make_plan(H, W, n_segments, ctypes.byref(p)))
    return p
