"""ctypes binding of libmckernels.so (include/mc_kernels.h).

Thin by design: every function takes raw device pointers (ints) and sizes, exactly as the
C-ABI declares them.  Used by the GPU parity tests, bench.py's per-kernel roofline leg and
__graft_entry__.smoke(); the training product path is the C++ runtime in libkernel.so, which
links the same library.  There is no CPU fallback: if the library is missing this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MC_KERNELS_LIB") or os.path.join(_HERE, "libmckernels.so")   # the override: tuning variants (tools/build_variant.sh)

c_fp = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_sz = ctypes.c_size_t
c_u64 = ctypes.c_ulonglong

_CONV_ARGS = [c_i] * 12 + [c_f, c_f, c_fp]          # N, C, imgY, imgX, F, fs, pad, stride, mY, mX, groups, (ps) ...

_SIGS = {
    "mc_conv_filter_acts": [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_f, c_f, c_fp],
    "mc_conv_filter_acts_fused": [c_fp, c_fp, c_fp, c_fp, c_i] + [c_i] * 11 + [c_f, c_f, c_fp],
    "mc_local_filter_acts": [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_f, c_f, c_fp],
    "mc_conv_img_acts": [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_f, c_f, c_fp],
    "mc_local_img_acts": [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_f, c_f, c_fp],
    "mc_conv_weight_acts": [c_fp, c_fp, c_fp] + [c_i] * 12 + [c_f, c_f, c_fp],
    "mc_local_weight_acts": [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_f, c_f, c_fp],
    "mc_conv_weight_acts_pick_partial_sum": [c_i] * 6,
    "mc_gemm": [c_fp, c_fp, c_fp] + [c_i] * 8 + [c_f, c_f, c_fp, c_i, c_fp, c_sz, c_fp],
    "mc_conv_weight_grad": [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_f, c_f, c_fp, c_sz, c_fp],
    "mc_conv_weight_grad_noalloc": [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_f, c_f, c_fp],
    "mc_conv_weight_grad_bias": [c_fp, c_fp, c_fp, c_fp] + [c_i] * 11 + [c_f, c_f, c_f, c_fp, c_sz, c_fp],
    "mc_gemm_row_sums": [c_fp, c_fp, c_fp] + [c_i] * 6 + [c_f, c_f, c_fp, c_f, c_fp, c_sz, c_fp],
    "mc_reserve_conv_split_workspace": [c_sz, c_fp],
    "mc_accumulate_f64": [c_fp, c_fp, c_i, c_fp],
    "mc_softmax_logreg": [c_fp] * 7 + [c_i, c_f, c_i, c_i, c_fp],
    "mc_set_autotune": [c_i],
    "mc_set_conv_band": [c_i],
    "mc_conv_filter_acts_relu_bits": [c_fp] * 5 + [c_i] * 10 + [c_fp],
    "mc_gemm_relu_bits": [c_fp, c_fp, c_fp] + [c_i] * 8 + [c_fp, c_fp, c_fp],
    "mc_gemm_relu_grad_bits": [c_fp, c_fp, c_fp] + [c_i] * 8 + [c_f, c_fp, c_fp],
    "mc_relu_bits_from_acts": [c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "mc_gemm_relu_grad": [c_fp, c_fp, c_fp] + [c_i] * 8 + [c_f, c_f, c_fp, c_i, c_fp, c_sz, c_fp],
    "mc_conv_local_max_undo_relu_grad": [c_fp] * 4 + [c_i] * 7 + [c_f, c_f, c_fp],
    "mc_conv_local_pool": [c_fp, c_fp] + [c_i] * 8 + [c_fp],
    "mc_conv_local_max_undo": [c_fp] * 4 + [c_i] * 7 + [c_f, c_f, c_fp],
    "mc_conv_local_pool_relu": [c_fp, c_fp, c_fp] + [c_i] * 8 + [c_fp],
    "mc_conv_local_max_undo_relu_out": [c_fp] * 4 + [c_i] * 7 + [c_f, c_f, c_i, c_fp],
    "mc_conv_local_pool_dropout": [c_fp] * 4 + [c_i] * 8 + [c_f, c_u64, c_u64, c_fp],
    "mc_conv_local_max_undo_dropout": [c_fp] * 5 + [c_i] * 7 + [c_f, c_f, c_i, c_fp],
    "mc_conv_local_avg_undo": [c_fp] * 2 + [c_i] * 7 + [c_f, c_f, c_fp],
    "mc_conv_local_avg_undo_relu_grad": [c_fp] * 3 + [c_i] * 7 + [c_f, c_fp],
    "mc_conv_response_norm": [c_fp] * 3 + [c_i] * 4 + [c_f, c_f, c_fp],
    "mc_conv_contrast_norm": [c_fp] * 4 + [c_i] * 4 + [c_f, c_f, c_fp],
    "mc_conv_response_norm_undo": [c_fp] * 5 + [c_i] * 4 + [c_f] * 4 + [c_fp],
    "mc_conv_response_norm_cross_map": [c_fp] * 3 + [c_i] * 4 + [c_f, c_f, c_i, c_f, c_fp],
    "mc_conv_response_norm_cross_map_undo": [c_fp] * 5 + [c_i] * 4 + [c_f, c_f, c_i, c_f, c_f, c_fp],
    "mc_conv_bed_of_nails": [c_fp, c_fp] + [c_i] * 5 + [c_f, c_f, c_fp],
    "mc_conv_bed_of_nails_undo": [c_fp, c_fp] + [c_i] * 5 + [c_f, c_f, c_fp],
    "mc_conv_gaussian_blur": [c_fp, c_fp, c_fp] + [c_i] * 5 + [c_f, c_f, c_fp],
    "mc_conv_resize_bilinear": [c_fp, c_fp] + [c_i] * 4 + [c_f, c_fp],
    "mc_conv_rgb_to_yuv": [c_fp, c_fp, c_i, c_i, c_fp],
    "mc_conv_rgb_to_lab": [c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "mc_eltwise_max": [c_fp, c_fp, c_fp, c_sz, c_fp],
    "mc_compute_eltwise_max_grad": [c_fp] * 4 + [c_i, c_sz, c_fp],
    "mc_conv_gather_channels": [c_fp, c_fp, c_fp] + [c_i] * 4 + [c_fp],
    "mc_conv_scatter_channels": [c_fp, c_fp, c_fp] + [c_i] * 4 + [c_f, c_f, c_fp],
    "mc_group_sparsity_in_label_cost": [c_fp] * 6 + [c_i] * 3 + [c_fp],
    "mc_group_sparsity_in_label_grad": [c_fp] * 4 + [c_i] * 3 + [c_f, c_f, c_fp],
    "mc_neuron_activate": [c_i, c_f, c_f, c_fp, c_fp, c_sz, c_fp],
    "mc_neuron_input_grad": [c_i, c_f, c_f, c_fp, c_fp, c_fp, c_i, c_sz, c_fp],
    "mc_dropout_apply": [c_fp, c_fp, c_fp, c_f, c_i, c_sz, c_fp],
    "mc_rand_uniform": [c_fp, c_sz, c_u64, c_u64, c_fp],
    "mc_dropout_counter": [c_fp, c_fp, c_f, c_i, c_sz, c_u64, c_u64, c_fp],
    "mc_add_row_vector": [c_fp, c_fp, c_i, c_i, c_f, c_fp],
    "mc_sum_rows": [c_fp, c_fp, c_i, c_i, c_f, c_f, c_fp],
    "mc_sum_cols": [c_fp, c_fp, c_i, c_i, c_f, c_f, c_fp],
    "mc_axpby": [c_fp, c_fp, c_f, c_f, c_sz, c_fp],
    "mc_scale": [c_fp, c_fp, c_f, c_sz, c_fp],
    "mc_sum_all": [c_fp, c_sz, c_fp, c_fp],
    "mc_softmax": [c_fp, c_fp, c_i, c_i, c_fp],
    "mc_compute_logreg_cost": [c_fp] * 4 + [c_i, c_i, c_fp],
    "mc_compute_logreg_cost_sums": [c_fp] * 5 + [c_i, c_i, c_fp],
    "mc_compute_logreg_grad": [c_fp] * 3 + [c_i, c_f, c_i, c_i, c_fp],
    "mc_compute_softmax_grad": [c_fp] * 3 + [c_i, c_i, c_i, c_fp],
    "mc_compute_logreg_softmax_grad": [c_fp] * 3 + [c_i, c_f, c_i, c_i, c_fp],
    "mc_weights_update": [c_fp] * 3 + [c_sz, c_f, c_f, c_f, c_fp],
    "mc_weights_update_spans": [c_fp] * 4 + [c_i, c_fp],
    "mc_describe_conv_filter_acts": [ctypes.c_char_p, c_i, c_i] + [c_i] * 11,
    "mc_device_info": [ctypes.c_char_p, c_i, ctypes.POINTER(c_i)],
}

NEURONS = {"ident": 0, "relu": 1, "brelu": 2, "abs": 3, "logistic": 4, "tanh": 5, "softrelu": 6,
           "square": 7, "sqrt": 8, "linear": 9}
POOL_MAX, POOL_AVG = 0, 1
NEURON_NONE, NEURON_RELU = 0, 1

_lib = None


class McError(RuntimeError):
    pass


def exported_symbols():
    """Names every build of the library must export (those include/mc_kernels.h declares)."""
    return sorted(list(_SIGS) + ["mc_last_error", "mc_gemm_workspace_bytes", "mc_conv_weight_grad_workspace_bytes",
                                   "mc_noalloc_scratch_bytes", "mc_conv_split_workspace_bytes", "mc_relu_bits_words",
                                   "mc_conv_weight_grad_bias_workspace_bytes", "mc_gemm_row_sums_workspace_bytes", "mc_dropout_keep_bytes"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise McError("libmckernels.so is not built (run `make -C cuda-convnet_amd` or "
                          "`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        L = ctypes.CDLL(_LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = c_i
        L.mc_last_error.restype = ctypes.c_char_p
        L.mc_gemm_workspace_bytes.argtypes = [c_i, c_i, c_i]
        L.mc_gemm_workspace_bytes.restype = c_sz
        L.mc_conv_weight_grad_workspace_bytes.argtypes = [c_i] * 6
        L.mc_conv_weight_grad_workspace_bytes.restype = c_sz
        L.mc_conv_weight_grad_bias_workspace_bytes.argtypes = [c_i] * 6
        L.mc_conv_weight_grad_bias_workspace_bytes.restype = c_sz
        L.mc_gemm_row_sums_workspace_bytes.argtypes = [c_i, c_i, c_i]
        L.mc_gemm_row_sums_workspace_bytes.restype = c_sz
        L.mc_noalloc_scratch_bytes.argtypes = []
        L.mc_noalloc_scratch_bytes.restype = c_sz
        L.mc_relu_bits_words.argtypes = [c_i, c_i]
        L.mc_relu_bits_words.restype = c_sz
        L.mc_conv_split_workspace_bytes.argtypes = [c_i] * 9
        L.mc_conv_split_workspace_bytes.restype = c_sz
        L.mc_dropout_keep_bytes.argtypes = [c_sz]
        L.mc_dropout_keep_bytes.restype = c_sz
        _lib = L
    return _lib


def call(name, *args):
    """Invoke an entry point; raise McError with the library's message on a rejected call."""
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise McError("%s failed (%d): %s" % (name, rc, L.mc_last_error().decode()))
    return rc


def device_info():
    buf = ctypes.create_string_buffer(64)
    n = c_i(0)
    call("mc_device_info", buf, 64, ctypes.byref(n))
    return buf.value.decode(), n.value
