"""ctypes binding of libskghoi_hip.so (include/skghoi.h).  The product has no CPU fallback: `lib()` raises if the
shared library has not been built (python __graft_entry__.py build / make -C skghoi_amd/csrc)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKG_LIB") or os.path.join(_HERE, "csrc", "libskghoi_hip.so")   # SKG_LIB: kernel A/B builds

EPI_BIAS, EPI_BIAS_RELU, EPI_MUL_RELU, EPI_RELU_DOT, EPI_BIAS_RES_RELU = range(5)
MAX_DET_PER_IMAGE = 1024
MAX_NODES = 160
SPATIAL_LD = 48
TRANSH_DIM = 50
TRANSH_ENT = 80
ABI_VERSION = 18
GEMM_GROUP_MAX = 4
CHECKSUM_PARTIALS = 1024
LOSS_CHUNKS = 64

_vp = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float


class GemmDesc(C.Structure):
    """Mirror of skg_gemm_desc."""
    _fields_ = [("A", _vp), ("lda", _i64), ("W", _vp), ("ldw", _i64), ("bias", _vp), ("C", _vp), ("ldc", _i64),
                ("M", _i32), ("N", _i32), ("K", _i32), ("epilogue", _i32), ("a_rows", _vp), ("out_rows", _vp),
                ("P", _vp), ("p_idx", _vp), ("ldp", _i64), ("Q", _vp), ("q_idx", _vp), ("ldq", _i64),
                ("mbias", _vp), ("C_raw", _vp), ("ldc_raw", _i64), ("dot_w", _vp), ("dot_partial", _vp),
                ("res", _vp), ("ldres", _i64), ("split_k", _i32), ("w_scale", _f32), ("split_ws", _vp), ("w_split", _vp),
                ("a_exp", _vp)]


class GemmXDesc(C.Structure):
    """Mirror of skg_gemmx_desc."""
    _fields_ = [("A", _vp), ("a_sm", _i64), ("a_sk", _i64), ("B", _vp), ("b_sn", _i64), ("b_sk", _i64),
                ("b_kshift", _i32), ("b_nshift", _i32), ("b_kstride", _i64), ("b_nstride", _i64),
                ("C", _vp), ("ldc", _i64), ("c_nshift", _i32), ("accumulate", _i32), ("c_nstride", _i64),
                ("M", _i32), ("N", _i32), ("K", _i32), ("relu", _i32), ("bias", _vp), ("mask", _vp), ("ldmask", _i64),
                ("a_rowsum", _vp), ("split_k", _i32), ("reserved", _i32), ("split_ws", _vp),
                ("A16", _vp), ("B16", _vp), ("C16", _vp), ("split_ctr", _vp),
                ("a16_ld", _i64), ("b16_ld", _i64)]


GEMMX_GROUP_MAX = 8
ADAMW_CHUNK = 16384


class GemmBf16Desc(C.Structure):
    """Mirror of skg_gemm_bf16_desc."""
    _fields_ = [("A", _vp), ("lda", _i64), ("W", _vp), ("ldw", _i64), ("bias", _vp), ("C", _vp), ("ldc", _i64),
                ("M", _i32), ("N", _i32), ("K", _i32), ("relu", _i32), ("out_bf16", _i32), ("split_k", _i32),
                ("split_ws", _vp)]


TRAIN_SEGS = ("W1_0", "W1_1", "W1_2", "W1_3", "b1_0", "b1_1", "b1_2", "b1_3", "W3_0", "W3_1", "W3_2", "W3_3", "b3", "W2",
              "b2", "clsW", "clsb", "nh_w", "nh_b", "no_w", "no_b", "adj_w", "adj_b", "sp0_w", "sp0_b", "sp2_w", "sp2_b",
              "sp4_w", "sp4_b", "fh_w", "fh_b", "ft_w", "ft_b", "bh3_w", "bh3_b", "bh1_w", "bh1_b")    # SKG_SEG_* order
TRAIN_BWD_STAGES = 12
COMM_ID_BYTES = 128


class Exchange(C.Structure):
    """Mirror of skg_exchange: the arena chunks a staged backward all-reduces itself over an skg_comm."""
    _fields_ = [("comm", C.c_void_p), ("arena", C.c_void_p), ("n_chunks", C.c_int32),
                ("stage", C.c_int32 * TRAIN_BWD_STAGES), ("end", C.c_int64 * TRAIN_BWD_STAGES),
                # the optimizer inside the backward (adamw = NULL: none): AdamW table entries per chunk + skg_adamw_f32's factors
                ("adamw", C.c_void_p), ("adamw_first", C.c_int32 * (TRAIN_BWD_STAGES + 1)), ("adamw_n_steps", C.c_int32),
                ("adamw_steps", C.c_void_p)] + [(n, C.c_double) for n in ("lr", "beta1", "beta2", "eps", "weight_decay",
                                                                          "bias1", "bias2")]


class Tuning(C.Structure):
    """Mirror of skg_tuning: the eval GEMM's developer switches, kept in a context (0 = the library's default)."""
    _fields_ = [(n, _i32) for n in ("small_mode", "small_tiles", "route_tiles", "khalves_blocks")]


class TrainPlan(C.Structure):
    """Mirror of skg_train_plan."""
    _fields_ = [(n, _i32) for n in ("NA", "Mg", "Mp", "Mh", "Mn", "A", "K", "Bf", "Cf", "x0_k", "bf16", "ld_logits")] + \
               [("params", _vp), ("grads", _vp), ("seg_off", _i64 * len(TRAIN_SEGS)),
                ("x0", _vp), ("gfeat", _vp), ("sp48", _vp), ("ent", _vp), ("meta", _vp)] + \
               [(n, _vp) for n in ("enc_row_hn", "img_hn", "ent_row_hn", "hum_img", "node_img", "grid_h", "grid_o",
                                   "grid_pair", "grid_img", "pair_grid", "pair_h", "pair_o", "pair_img", "hum_of",
                                   "node_of")] + \
               [("ws", _vp), ("ws_floats", _i64), ("pair_features", _vp), ("logits", _vp), ("dlogits", _vp),
                ("dx0", _vp), ("dgfeat", _vp), ("timer", _vp), ("ws16", _vp), ("params16", _vp), ("pf16", _vp),
                ("params_floats", _i64), ("counters", _vp), ("n_counters", _i64), ("split_target", _i32), ("split_max", _i32),
                ("two_branch", _i32), ("reserved2", _i32)]


LAYOUT_SLICES = ("meta", "node_img", "hum_img", "node_enc_row", "hum_enc_row", "node_ent_row", "hum_ent_row", "enc_row_hn",
                 "img_hn", "ent_row_hn", "hum_of", "node_of", "pair_img", "gt_off", "active")       # SKG_LAY_* order


class LayoutInfo(C.Structure):
    """Mirror of skg_layout_info."""
    _fields_ = [(n, _i32) for n in ("B", "n_visit", "n_active", "index_error")] + \
               [(n, _i64) for n in ("sum_all", "sum_n", "sum_h", "sum_g", "sum_p", "sum_l", "ints")] + \
               [("off", _i32 * len(LAYOUT_SLICES)), ("len", _i32 * len(LAYOUT_SLICES))]


# numpy dtype of skg_image_meta (12 x 4 bytes)
META_FIELDS = [("image", "i4"), ("n_h", "i4"), ("n", "i4"), ("box_off", "i4"), ("enc_off", "i4"), ("node_off", "i4"),
               ("hum_off", "i4"), ("grid_off", "i4"), ("pair_off", "i4"), ("out_off", "i4"), ("img_h", "f4"),
               ("img_w", "f4")]

PROTOTYPES = {
    "skg_abi_version": (C.c_int, []),
    "skg_build_info": (C.c_char_p, []),
    "skg_preprocess_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _f32, C.c_int, C.c_int, _vp, C.c_int,
                                     _f32, _vp, _vp, _vp]),
    "skg_pack_detections_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "skg_pairs_spatial_f32": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int,
                                        _vp]),
    "skg_pairs_spatial_padded_f32": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                               C.c_int, C.c_int, C.c_int, _vp]),
    "skg_roi_align_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32, C.c_int, _vp, _vp,
                                    C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "skg_roi_align_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32, C.c_int, _vp, _vp,
                                    C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "skg_global_avgpool_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "skg_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "skg_transh_draw_f32": (C.c_int, [_vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "skg_transh_draw_train_f32": (C.c_int, [_vp, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "skg_gemm_dot_partials": (C.c_int, [C.POINTER(GemmDesc)]),
    "skg_split_weights_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "skg_split_weights_f16x2": (C.c_int, [_vp, C.c_int, C.c_int, _i64, _f32, _vp, _vp]),
    "skg_gemm_group_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_int, _vp]),
    "skg_gemm_group_tile": (C.c_int, [C.POINTER(GemmDesc), C.c_int]),
    "skg_row_exponents_f32": (C.c_int, [_vp, C.c_int64, _vp, C.c_int, C.c_int, _vp, _vp]),
    "skg_adamw_f32": (C.c_int, [_vp, C.c_int] + [C.c_double] * 7 + [_vp, C.c_int, _vp]),
    "skg_ctx_set_tuning": (C.c_int, [_vp, C.POINTER(Tuning)]),
    "skg_ctx_get_tuning": (C.c_int, [_vp, C.POINTER(Tuning)]),
    "skg_ctx_make_current": (_vp, [_vp]),
    "skg_gemmx_ws_floats": (C.c_int64, [C.POINTER(GemmXDesc)]),
    "skg_gemmx_f32": (C.c_int, [C.POINTER(GemmXDesc), C.c_int, _vp]),
    "skg_gemmx_bf16": (C.c_int, [C.POINTER(GemmXDesc), C.c_int, _vp]),
    "skg_gemmx_path_counts": (None, [C.POINTER(_i64), C.c_int]),
    "skg_gemm_bf16": (C.c_int, [C.POINTER(GemmBf16Desc), _vp]),
    "skg_transpose_bf16": (C.c_int, [_vp, _i64, C.c_int, C.c_int, _vp, _i64, _vp]),
    "skg_transpose_f32": (C.c_int, [_vp, _i64, C.c_int, C.c_int, _vp, _i64, _vp]),
    "skg_concat_entity_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, C.c_int, _vp, _i64, _vp]),
    "skg_rows_mul_relu_f32": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp,
                                        _i64, _vp]),
    "skg_graph_aggregate_f32": (C.c_int, [_vp, C.c_int, _i64, _f32, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp,
                                          _vp, _i64, C.c_int, _vp, _vp, _i64, _vp, _vp]),
    "skg_layernorm_f32": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, C.c_int, _f32, _vp, _i64, _vp]),
    "skg_layernorm2_f32": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, _vp, _i64, _vp, _i64, _vp, _vp, C.c_int, _vp, _i64,
                                     C.c_int, _f32, _vp]),
    "skg_postprocess_f32": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int,
                                      _f32, _i64, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "skg_associate_f32": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _f32, _vp, _vp, _vp]),
    "skg_transh_scores_f32": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "skg_param_checksum": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "skg_graph_aggregate_train_f32": (C.c_int, [_vp, C.c_int, _i64, _f32, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp,
                                                _vp, _i64, C.c_int, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "skg_rowdot_f32": (C.c_int, [_vp, _i64, _vp, C.c_int, C.c_int, _vp, _vp]),
    "skg_add_layernorm_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, C.c_int, _f32, _vp, _vp, _vp, _vp]),
    "skg_layernorm_bwd_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "skg_mul_bwd_f32": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, C.c_int, _vp, _i64,
                                  C.c_int, _vp]),
    "skg_segment_sum_f32": (C.c_int, [_vp, _i64, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int,
                                      _vp]),
    "skg_aggregate_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "skg_adjacency_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "skg_entity_rows_bwd_f32": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "skg_eval_associate_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp,
                                         _vp, _vp, _f32, _vp, _vp, _vp, _vp]),
    "skg_eval_ap11_f64": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "skg_transh_sample_ws_ints": (C.c_int64, [C.c_int, C.c_int]),
    "skg_transh_sample_f32": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _f32, _vp, _vp, _vp, _vp, _vp,
                                        _vp]),
    "skg_hoi_loss_f32": (C.c_int, [_vp, _i64, C.c_int, _vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "skg_count_positives_f32": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _f32, _vp, _vp]),
    "skg_loss_finish_f32": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _i64, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "skg_scale_dlogits_f32": (C.c_int, [_vp, _i64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "skg_train_ws_floats": (C.c_int64, [C.POINTER(TrainPlan)]),
    "skg_train_forward_f32": (C.c_int, [C.POINTER(TrainPlan), C.c_int, _vp]),
    "skg_train_backward_f32": (C.c_int, [C.POINTER(TrainPlan), C.c_int, C.c_int, _vp]),
    "skg_train_backward_async_f32": (C.c_int, [C.POINTER(TrainPlan), C.c_int, C.c_int, _vp]),
    "skg_train_backward_join": (C.c_int, []),
    "skg_twin_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "skg_layout_pack_train": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _i64,
                                        C.POINTER(LayoutInfo)]),
    "skg_train_timer_create": (_vp, [C.c_int]),
    "skg_train_timer_destroy": (None, [_vp]),
    "skg_train_timer_read": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "skg_context_create": (_vp, []),
    "skg_context_destroy": (None, [_vp]),
    "skg_ctx_train_backward_async_f32": (C.c_int, [_vp, C.POINTER(TrainPlan), C.c_int, C.c_int, _vp, _vp, C.c_uint32]),
    "skg_ctx_train_forward_f32": (C.c_int, [_vp, C.POINTER(TrainPlan), C.c_int, _vp]),
    "skg_ctx_train_backward_exchange_f32": (C.c_int, [_vp, C.POINTER(TrainPlan), C.c_int, C.c_int, _vp, _vp, C.POINTER(Exchange)]),
    "skg_sizeof_exchange": (C.c_int, []),
    "skg_comm_load": (C.c_int, [C.c_char_p]),
    "skg_comm_unique_id": (C.c_int, [_vp]),
    "skg_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "skg_comm_destroy": (None, [_vp]),
    "skg_comm_abort": (C.c_int, [_vp]),
    "skg_comm_dead": (C.c_int, [_vp]),
    "skg_comm_world": (C.c_int, [_vp]),
    "skg_comm_rank": (C.c_int, [_vp]),
    "skg_comm_collectives": (_i64, [_vp]),
    "skg_comm_last_error": (C.c_char_p, []),
    "skg_comm_all_reduce_chunks_f32": (C.c_int, [_vp, _vp, C.POINTER(_i64), C.c_int, _vp]),
    "skg_comm_all_reduce_begin_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "skg_comm_all_reduce_end": (C.c_int, [_vp, _vp]),
    "skg_comm_exposed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "skg_ctx_stream_wait_stage": (C.c_int, [_vp, C.c_int, _vp]),
    "skg_ctx_train_backward_stage_wait": (C.c_int, [_vp, C.c_int]),
    "skg_ctx_train_backward_join": (C.c_int, [_vp]),
    "skg_ctx_train_backward_progress": (C.c_int, [_vp]),
    "skg_train_ws_offset": (C.c_int64, [C.POINTER(TrainPlan), C.c_int]),
    "skg_train_flops": (C.c_double, [C.POINTER(TrainPlan), C.c_int]),
}

_LIB = None


class SkgError(RuntimeError):
    pass


def lib():
    """Loads libskghoi_hip.so once; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.isfile(LIB_PATH):
        raise SkgError("libskghoi_hip.so not built at %s -- run `python __graft_entry__.py` or "
                       "`make -C skghoi_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    if l.skg_abi_version() == ABI_VERSION and l.skg_sizeof_exchange() != C.sizeof(Exchange):
        raise SkgError("skg_exchange is %d bytes in libskghoi_hip.so, %d in the binding" % (l.skg_sizeof_exchange(), C.sizeof(Exchange)))
    if l.skg_abi_version() != ABI_VERSION:
        raise SkgError("libskghoi_hip.so ABI %d != binding ABI %d" % (l.skg_abi_version(), ABI_VERSION))
    _LIB = l
    # developer switches of the eval GEMM from the environment (SKG_SMALL_MODE: 64 x 64 main loop of the small launches,
    # SKG_SMALL_TILES: bound of the 64 x 64-tile launches, SKG_ROUTE_TILES: mid-size launches on the free-layout GEMM from
    # here up, SKG_KHALVES_BLOCKS): they go into a context of THIS module, current on the loading thread -- the library
    # itself has no process-wide switch any more
    env = {k: int(os.environ[e]) for k, e in (("small_mode", "SKG_SMALL_MODE"), ("small_tiles", "SKG_SMALL_TILES"),
                                              ("route_tiles", "SKG_ROUTE_TILES"), ("khalves_blocks", "SKG_KHALVES_BLOCKS"))
           if os.environ.get(e)}
    if env:
        set_tuning(**env)
    return l


_TUNE_CTX = None


def set_tuning(**kw):
    """Developer switches of the eval GEMM (skg_tuning fields; 0 = the library's default) for the CALLING THREAD: they are
    written to a context owned by this module, which is made the thread's current context.  Returns the previous values, so
    that `old = set_tuning(route_tiles=1 << 30); ...; set_tuning(**old)` restores them."""
    global _TUNE_CTX
    l = lib()
    if _TUNE_CTX is None:
        _TUNE_CTX = l.skg_context_create()
        if not _TUNE_CTX:
            raise SkgError("skg_context_create failed")
    t = Tuning()
    check(l.skg_ctx_get_tuning(_TUNE_CTX, C.byref(t)), "skg_ctx_get_tuning")
    old = {n: int(getattr(t, n)) for n, _ in Tuning._fields_}
    for k, v in kw.items():
        if k not in old:
            raise TypeError("unknown tuning switch %r" % k)
        setattr(t, k, int(v))
    check(l.skg_ctx_set_tuning(_TUNE_CTX, C.byref(t)), "skg_ctx_set_tuning")
    l.skg_ctx_make_current(_TUNE_CTX)
    return old


_ERR = {-1: "SKG_E_ARG (bad argument)", -2: "SKG_E_ALIGN (pointer / leading dimension not 16-byte aligned)",
        -3: "SKG_E_LIMIT (compiled-in limit exceeded)", -4: "SKG_E_UNSUPPORTED (RCCL not available in this process)",
        -5: "SKG_E_COMM (an RCCL call failed)"}


def check(rc, what):
    if rc != 0:
        detail = ""
        if rc in (-4, -5):
            detail = " -- " + (lib().skg_comm_last_error() or b"").decode(errors="replace")
        raise SkgError("%s failed: %s%s" % (what, _ERR.get(rc, "hipError_t %d" % rc), detail))
