"""ctypes binding of libkbner_hip.so (the C ABI declared in include/kbner.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises.
torch is used only for device memory and streams (tensor.data_ptr(), current_stream().cuda_stream).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkbner_hip.so")   # (lab tools point this at experiment builds before load(): tools/labenv.py)

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
U32 = ctypes.c_uint32
P = c_void_p

# name -> (restype, argtypes) ; must list EVERY symbol include/kbner.h declares
SIGNATURES = {
    "kbner_abi_version": (c_int, []),
    "kbner_device_count": (c_int, []),
    "kbner_crf_viterbi_lds_bytes": (c_size_t, [c_int, c_int]),
    "kbner_crf_viterbi": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    "kbner_crf_nll_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    "kbner_crf_nll_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "kbner_crf_posterior": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "kbner_crf_posterior_kl_ws_floats": (c_size_t, [c_int, c_int, c_int]),
    "kbner_crf_posterior_kl": (c_int, [P, P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    "kbner_crf_fb_score": (c_int, [P, P, P, U32, c_int, c_int, c_int, c_int, c_int, P, P]),
    "kbner_crf_posterior_kl_scores": (c_int, [P, P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    "kbner_crf_pair_ws_floats": (c_size_t, [c_int, c_int, c_int]),
    "kbner_crf_pair_posterior": (c_int, [P, P, P, U32, c_float, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    "kbner_emission_kl": (c_int, [P, P, P, P, c_float, c_int, c_int, c_int, c_int, P, P, P]),
    "kbner_softmax_ce": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "kbner_softmax_decode": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P]),
    "kbner_crf_exact_kd": (c_int, [P, P, P, P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    "kbner_crf_viterbi_nbest_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "kbner_crf_viterbi_nbest": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    "kbner_gather_rows": (c_int, [P, P, P, c_int, c_int, P]),
    "kbner_gather_rows_ld": (c_int, [P, c_int, P, P, c_int, c_int, c_int, P]),
    "kbner_gather_rows_f32": (c_int, [P, P, P, c_int, c_int, P]),
    "kbner_l2_rows": (c_int, [P, P, P, c_float, P, P, c_int, c_int, P]),
    "kbner_scatter_add_rows_f32": (c_int, [P, P, P, c_int, c_int, P]),
    "kbner_scatter_rows": (c_int, [P, P, P, c_int, c_int, P]),
    "kbner_scatter_rows_f32": (c_int, [P, P, P, c_int, c_int, P]),
    "kbner_head_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "kbner_head_bwd_dx": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "kbner_head_bwd_dw": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "kbner_colsum": (c_int, [P, P, c_int, c_int, c_int, P]),
    "kbner_ln_fwd": (c_int, [P, P, P, c_float, P, P, P, c_int, c_int, P]),
    "kbner_ln_bwd_ws_floats": (c_int, [c_int]),
    "kbner_ln_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, P, U32, U32, P]),
    "kbner_embed_ln_fwd": (c_int, [P, P, P, P, P, P, P, c_float, P, P, P, P, c_int, c_int, U32, U32, P]),
    "kbner_embed_ln_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, U32, U32, P]),
    "kbner_embed_ln_bwd_mark": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, U32, U32, P]),
    "kbner_dropout_mask": (c_int, [P, c_int, c_int, c_int, U32, U32, P]),
    "kbner_gemm_bf16": (c_int, [c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, P, c_int, P, c_int,
                                P, c_int, c_int, c_int, c_float, U32, U32, P]),
    "kbner_gemm_bf16_grouped": (c_int, [c_int, c_int, P, P]),
    "kbner_colsum_rows_f32": (c_int, [P, c_int, c_int, P, P]),
    "kbner_colsum_rows_f32_batched": (c_int, [P, c_int, c_int, P]),
    "kbner_ln_bwd_blocks": (c_int, [c_int]),
    "kbner_ln_fwd_slabs": (c_int, [P, c_int, P, P, c_int, U32, U32, P, P, P, c_float, P, P, P, c_int, c_int, P]),
    "kbner_ln_bwd_slabs": (c_int, [P, c_int, P, c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, P, U32, U32, P]),
    "kbner_ln_colreduce_batched": (c_int, [P, c_int, c_int, P]),
    "kbner_gemm_tile_rows": (c_int, [c_int, c_int, c_int]),
    "kbner_gemm_bf16_grouped_dyn": (c_int, [c_int, c_int, P, P, P]),
    "kbner_gemm_set_variant": (c_int, [c_int]),
    "kbner_gemm_get_variant": (c_int, []),
    "kbner_splitk_finish": (c_int, [P, c_int, P, P, c_int, P, c_int, c_int, c_int, U32, U32, P]),
    "kbner_attn_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, U32, U32, P]),
    "kbner_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, U32, U32, P, P]),
    "kbner_lstm_step": (c_int, [P, c_int, P, P, P, P, P, P, c_int, c_int, P, c_int, c_int, c_int, P]),
    "kbner_lstm_seq": (c_int, [P, c_int, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "kbner_sqnorm_ws_floats": (c_int, []),
    "kbner_grad_sqnorm": (c_int, [P, c_size_t, P, P, c_int, P]),
    "kbner_mark_rows": (c_int, [P, c_int, P, c_int, P]),
    "kbner_grad_sqnorm_rows": (c_int, [P, P, c_int, c_int, P, P, c_int, P]),
    "kbner_adamw_hf_rows": (c_int, [P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, c_float, P, c_float, c_float, c_int, P]),
    "kbner_adamw_hf_rows_lazy": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, c_float,
                                          c_float, P]),
    "kbner_adamw_rows_catchup": (c_int, [P, c_int, P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_float, P]),
    "kbner_adamw_hf": (c_int, [P, P, P, P, P, c_size_t, c_size_t, c_float, c_float, c_float, c_float, c_float, P, c_float,
                               c_float, c_int, P]),
    "kbner_f32_to_bf16": (c_int, [P, P, c_size_t, P]),
    "kbner_bf16_to_f32": (c_int, [P, P, c_size_t, P]),
    "kbner_wdiff_sum": (c_int, [P, P, P, c_int, P, P]),
    "kbner_probe_tr": (c_int, [P, P, P]),
    "kbner_probe_mfma": (c_int, [P, P, P, P]),
}

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_GELU, EPI_ADD, EPI_DGELU, EPI_ATOMIC32, EPI_RMW32, EPI_COLSUM, EPI_DROP = 1, 2, 4, 8, 16, 32, 64, 128
EPI_STORE32 = 256
EPI_COLSUM_WS = 512
EPI_GELU_FWD = 1024


class GemmProblem(ctypes.Structure):
    """mirror of kbner_gemm_problem (include/kbner.h)"""
    _fields_ = [("A", P), ("B", P), ("C", P), ("C32", P), ("bias", P), ("addend", P), ("aux", P), ("out2", P), ("colsum", P),
                ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int), ("ldc", c_int), ("ldc32", c_int),
                ("ldadd", c_int), ("ldaux", c_int), ("ldout2", c_int), ("epi", c_int), ("alpha", c_float), ("drop_seed", U32),
                ("drop_thresh", U32)]

_lib = None


class KbnerError(RuntimeError):
    pass


def load():
    """Load the library and bind every symbol; raises if the .so or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KbnerError("libkbner_hip.so not built: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                         "There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


_stream_cached = None


def stream_ptr():
    """hipStream_t of torch's current stream (cached inside a stream_scope: the lookup costs ~3 us, an engine micro-batch
    makes ~450 launches)"""
    if _stream_cached is not None:
        return _stream_cached
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class stream_scope:
    """pins stream_ptr() to the stream that is current on entry; re-entrant"""

    def __enter__(self):
        global _stream_cached
        self._prev = _stream_cached
        if _stream_cached is None:
            import torch
            _stream_cached = c_void_p(torch.cuda.current_stream().cuda_stream)
        return self

    def __exit__(self, *exc):
        global _stream_cached
        _stream_cached = self._prev
        return False


def check(rc, what):
    if rc != 0:
        raise KbnerError("%s failed with code %d" % (what, rc))


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise KbnerError("%s failed with code %d" % (name, rc))
