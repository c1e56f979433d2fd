"""ctypes binding of libvaenmf.so (include/vaenmf.h).  There is NO fallback: if the
HIP library is missing or a call fails, this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAENMF_LIB") or os.path.join(_HERE, "libvaenmf.so")   # VAENMF_LIB: another build of the same library (dev A/B runs)

PREC_BF16X3, PREC_BF16 = 0, 1
RNG_REPLAY, RNG_DEVICE = 0, 1
Q_FS, Q_KP, Q_TILES, Q_NT, Q_NUTT, Q_MSTEP_PATH, Q_WTILES, Q_EM_GRAPH, Q_DEV_ALLOCS, Q_W_FUSED, Q_CHAIN_KERNEL = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
ACT_NONE, ACT_TANH, ACT_RELU, ACT_SIGMOID, ACT_STEP = 0, 1, 2, 3, 4
LABEL_IBM, LABEL_VAD = 0, 1


class Config(C.Structure):
    _fields_ = [("F", C.c_int32), ("K", C.c_int32), ("L", C.c_int32), ("H1", C.c_int32), ("H2", C.c_int32),
                ("max_frames", C.c_int32), ("max_utts", C.c_int32), ("precision", C.c_int32)]


class Rng(C.Structure):
    _fields_ = [("mode", C.c_int32), ("call", C.c_uint32), ("eps", C.c_void_p), ("u", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/vaenmf.h declares
_P, _I, _I64, _F, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
SIGNATURES = {
    "vaenmf_last_error": (C.c_char_p, []),
    "vaenmf_plan_create": (_I, [C.POINTER(Config), C.POINTER(_P)]),
    "vaenmf_plan_destroy": (None, [_P]),
    "vaenmf_plan_query": (_I, [_P, _I]),
    "vaenmf_set_decoder_weights": (_I, [_P, _P, _I, _P, _P, _P, _P, _P]),
    "vaenmf_bind_batch": (_I, [_P, _I, _P, _P]),
    "vaenmf_bind_batch_async": (_I, [_P, _I, _P, _P, _P]),
    "vaenmf_set_noise_psd": (_I, [_P, _P]),
    "vaenmf_init_nmf": (_I, [_P, _P, _P, _P, C.c_uint64, _F, _P]),
    "vaenmf_layer1_bias": (_I, [_P, _P, _I, _P, _P]),
    "vaenmf_mh_chain": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _F, C.POINTER(Rng), _P, _P]),
    "vaenmf_sample_store": (_I, [_P, _I]),
    "vaenmf_sample_store_gather": (_I, [_P, _P, _P]),
    "vaenmf_m_step_stored": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "vaenmf_wiener_stored": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vaenmf_rng_fill": (_I, [_P, C.c_uint32, _I, _P, _P, _P]),
    "vaenmf_decode": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "vaenmf_m_step": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "vaenmf_wiener": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vaenmf_em_run": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "vaenmf_dense": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P]),
    "vaenmf_power_spec": (_I, [_P, _P, _I64, _P]),
    "vaenmf_stft_num_frames": (_I, [_I64, _D, _D, _D, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "vaenmf_stft_batch": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "vaenmf_istft_batch": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "vaenmf_lorenz_work_bytes": (_I64, [_I, _I, _I, _I]),
    "vaenmf_lorenz_labels": (_I, [_P, _I, _P, _I, _I, _I, _F, _F, _F, _P, _I, _P, _P, _I64, _P]),
    "vaenmf_wiener_mask": (_I, [_P, _P, _I64, _F, _P, _P]),
    "vaenmf_apply_mask": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "vaenmf_spp_estimate": (_I, [_P, _I, _I, _P, _I, _D, _D, _D, _D, _I, _P, _P, _I, _P]),
    "vaenmf_spp_noise_given": (_I, [_P, _P, _I64, _D, _P, _P]),
    "vaenmf_gram3_batch": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "vaenmf_profile_enable": (_I, [_P, _I]),
    "vaenmf_profile_read": (_I, [_P, _P, _P]),
    "vaenmf_wchain_addressable": (_I, [_I64, _I, _I, _I, _I, _I, _I]),
    "vaenmf_hbm_read_probe": (_I, [_P, _I64, _I, _P, C.POINTER(_D), _P]),
}

_lib = None


def lib():
    """The loaded library; raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libvaenmf.so not found at %s -- build it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class VaenmfError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise VaenmfError("libvaenmf: %s (code %d)" % (lib().vaenmf_last_error().decode(), rc))
