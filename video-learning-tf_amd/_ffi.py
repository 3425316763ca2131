"""ctypes binding of libvltf_hip.so (include/vltf.h).  There is no CPU fallback: if the library is
missing or a call fails this module raises, loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLTF_HIP_LIB") or os.path.join(_HERE, "libvltf_hip.so")   # override: kernel experiments only

p = C.c_void_p
i32, i64, f32, u64, sz = C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); mirrors include/vltf.h one to one
SIGNATURES = {
    "vl_last_error": (C.c_char_p, []),
    "vl_version": (i32, []),
    "vl_device_count": (i32, []),
    "vl_input_prep_u8": (i32, [p, p, i32, i32, i32, i32, i32, p, p, p, p, i32, i32, p]),
    "vl_nhwc_to_nchw": (i32, [p, p, i32, i32, i32, i32, i32, i32, p]),
    "vl_nchw_to_nhwc": (i32, [p, p, i32, i32, i32, i32, p]),
    "vl_conv_create": (i32, [C.POINTER(p), i32, i32, i32, i32, i32, i32, i32, i32]),
    "vl_conv_destroy": (None, [p]),
    "vl_conv_out_hw": (i32, [p, C.POINTER(i32), C.POINTER(i32)]),
    "vl_conv_set_halo": (i32, [p, i32, i32, i32, i32]),
    "vl_conv_set_x_phase_split": (i32, [p, i32]),
    "vl_conv_x_phase": (i32, [p]),
    "vl_set_conv_math": (i32, [i32]),
    "vl_conv_math": (i32, []),
    "vl_gemm_split_ws_bytes": (C.c_size_t, [i32, i32, i32]),
    "vl_conv_fwd": (i32, [p, p, p, p, p, i32, i32, p]),
    "vl_conv_wt_transpose": (i32, [p, p, p, p]),
    "vl_conv_dgrad": (i32, [p, p, p, p, p, i32, p]),
    "vl_conv_wgrad_ws_bytes": (sz, [p, i32]),
    "vl_conv_wgrad_fuses_bias": (i32, [p]),
    "vl_conv_wgrad": (i32, [p, p, p, p, p, p, sz, i32, p]),
    "vl_c8_bytes": (sz, [i32, i32, i32, i32, i32]),
    "vl_pack_c8": (i32, [p, p, i32, i32, i32, i32, i32, i32, p]),
    "vl_conv_c8_w_bytes": (sz, [p, i32]),
    "vl_conv_c8_pack_w": (i32, [p, p, p, i32, p]),
    "vl_conv_c8_fwd": (i32, [p, p, p, p, p, p, i32, i32, p]),
    "vl_conv_c8_dgrad": (i32, [p, p, p, p, p, p, p, i32, p]),
    "vl_pack_kc8": (i32, [p, p, i64, i32, i64, i64, p]),
    "vl_gemm_kc8_ws_bytes": (sz, [i32, i32, i32]),
    "vl_gemm_kc8": (i32, [p, p, p, i32, i32, i32, p, i32, p, sz, p]),
    "vl_s2d_c8_from_x0": (i32, [p, p, p, i32, p]),
    "vl_input_prep_u8_s2d": (i32, [p, p, p, i32, i32, i32, p, p, p, p, p]),
    "vl_s2d_weights": (i32, [p, p, p, i32, p]),
    "vl_bias_grad_c8": (i32, [p, p, p, i32, i32, i32, i32, i32, p]),
    "vl_conv_c8_wgrad_ws_bytes": (sz, [p, i32]),
    "vl_conv_c8_wgrad": (i32, [p, p, p, p, p, sz, i32, p]),
    "vl_bias_grad_nchw": (i32, [p, p, p, i32, i32, i32, p]),
    "vl_lrn_fwd": (i32, [p, p, i32, i32, i32, i32, f32, f32, f32, p]),
    "vl_lrn_bwd": (i32, [p, p, p, i32, i32, i32, i32, f32, f32, f32, i32, i32, i32, p]),
    "vl_lrn_pool_fwd": (i32, [p, p, p, i32, i32, i32, i32, i32, i32, f32, f32, f32, p]),
    "vl_pool_lrn_bwd": (i32, [p, p, p, p, i32, i32, i32, i32, i32, i32, f32, f32, f32, i32, i32, p]),
    "vl_pool_lrn_bwd_test_ranges": (i32, [i32]),
    "vl_lrn_pool_fwd_c8": (i32, [p, i32, p, p, i32, i32, i32, i32, i32, i32, f32, f32, f32, p]),
    "vl_pool_lrn_bwd_c8": (i32, [p, i32, p, p, p, i32, i32, i32, i32, i32, i32, f32, f32, f32, i32, i32, p]),
    "vl_maxpool_fwd": (i32, [p, p, p, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, p]),
    "vl_maxpool_bwd": (i32, [p, p, p, p, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, i32, p]),
    "vl_gemm": (i32, [i32, i32, i32, i32, i32, p, i64, p, i64, p, i64, p, i32, p, p, sz, p]),
    "vl_colsum": (i32, [p, i64, p, p, i32, i32, p]),
    "vl_lstm_step_fwd": (i32, [p, p, p, p, p, p, i32, i32, i32, i32, f32, p]),
    "vl_lstm_step_bwd": (i32, [p, p, p, p, p, p, i32, i32, i32, i32, p]),
    "vl_lstm_seq_ws_bytes": (sz, [i32, i32, i32]),
    "vl_lstm_seq_fwd": (i32, [p, p, p, p, p, p, p, p, i32, i32, i32, f32, p, sz, p]),
    "vl_lstm_seq_bwd": (i32, [p, p, p, p, p, p, p, p, i32, i32, i32, p, sz, p]),
    "vl_lstm_seq_status": (i32, [p, C.POINTER(i32)]),
    "vl_lstm_seq_test_hooks": (i32, [C.c_uint, i32]),
    "vl_transpose": (i32, [p, i64, p, i32, i32, p]),
    "vl_temporal_fusion_fwd": (i32, [p, p, i32, i32, i32, i32, p]),
    "vl_temporal_fusion_bwd": (i32, [p, p, i32, i32, i32, i32, p]),
    "vl_dropout_fwd": (i32, [p, p, p, i64, f32, u64, p]),
    "vl_dropout_bwd": (i32, [p, p, p, i64, f32, p]),
    "vl_softmax_xent": (i32, [p, p, p, p, p, i32, i32, f32, p]),
    "vl_sumsq": (i32, [p, i64, p, p, i32, p]),
    "vl_sgd_apply": (i32, [p, p, i64, f32, f32, p, f32, p, p]),
    "vl_adam_apply": (i32, [p, p, p, p, i64, f32, f32, p, f32, i32, p, p]),
    "vl_status_or": (i32, [p, p, i32, p]),
    "vl_fill": (i32, [p, i64, f32, p]),
    "vl_resize_create": (i32, [C.POINTER(p), i32, i32, i32, i32, i32]),
    "vl_resize_destroy": (None, [p]),
    "vl_resize_tmp_bytes": (sz, [p, i32]),
    "vl_resize_u8": (i32, [p, p, p, p, i32, p]),
    "vl_copy2d": (i32, [p, i64, p, i64, i32, i32, p]),
    "vl_eltwise2": (i32, [p, p, p, i64, i32, p]),
    "vl_max2_grad": (i32, [p, p, p, p, p, i64, p]),
    "vl_fuse_n": (i32, [p, i32, p, i64, i32, p]),
    "vl_fuse_n_grad": (i32, [p, i32, p, p, i64, i32, p]),
    "vl_relu_grad": (i32, [p, p, i64, p]),
}


class VltfError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library; raises VltfError when it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VltfError("libvltf_hip.so not found at %s -- build it with "
                            "`python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)" % LIB_PATH)
        # ONE HIP runtime per process: torch bundles its own libamdhip64.so (soname libamdhip64.so.7, the
        # same as /opt/rocm's).  Loading torch first makes our NEEDED entry resolve to that copy; the other
        # order would put two runtimes in the process and device pointers would not be shared.
        import torch  # noqa: F401
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tl):
            C.CDLL(tl, mode=C.RTLD_GLOBAL)
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as ex:
            raise VltfError("cannot load %s: %s" % (LIB_PATH, ex))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                raise VltfError("libvltf_hip.so does not export %s (stale build?)" % name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().vl_last_error()
        raise VltfError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def call(name, *args):
    """Call an int-returning entry point and raise on failure."""
    check(getattr(lib(), name)(*args), name)
