"""ctypes binding of the C ABI in include/autoreparam.h.

The shared library is the product: if it is missing or fails to load this module
raises -- there is no CPU fallback on the product path (the CPU restatement under
oracle/ is test infrastructure and is never imported from here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARP_LIB_PATH: timing experiments only (a variant built with ARP_HIPCC_FLAGS / ARP_BUILD_TAG); honoured under
# ARP_DEBUG=1 only and announced on stderr
LIB_PATH = os.path.join(_HERE, "libautoreparam_hip.so")
if os.environ.get("ARP_LIB_PATH"):
    import sys as _sys
    if os.environ.get("ARP_DEBUG") == "1":
        LIB_PATH = os.environ["ARP_LIB_PATH"]
        print("autoreparam_amd: DEBUG SWITCH ARP_LIB_PATH=%s is in effect (ARP_DEBUG=1)" % LIB_PATH, file=_sys.stderr, flush=True)
    else:
        print("autoreparam_amd: ARP_LIB_PATH IGNORED (experiment switch; set ARP_DEBUG=1 to enable it)", file=_sys.stderr, flush=True)

MODEL_EIGHT_SCHOOLS, MODEL_RADON, MODEL_GERMAN_CREDIT, MODEL_ELECTION, MODEL_RADON_STDDVS = 0, 1, 2, 3, 4
MODEL_NEALS_FUNNEL = 5
MODEL_ELECTRIC = 6
MODEL_TIME_SERIES = 7
ADAPT_NONE, ADAPT_DUAL, ADAPT_SIMPLE = 0, 1, 2
RNG_SLOTS = 16  # rng buffer is [C][16][4] uint32

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


class Dataset(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_obs", C.c_int32), ("n_groups", C.c_int32),
                ("n_features", C.c_int32),
                ("group_host", _i32p), ("u_host", _f32p), ("x_host", _f32p), ("x2_host", _f32p),
                ("y_host", _f32p), ("X_host", _f32p), ("group2_host", _i32p), ("group3_host", _i32p)]


class HmcConfig(C.Structure):
    _fields_ = [("n_chains", C.c_int32), ("n_leapfrog", C.c_int32), ("n_steps", C.c_int32),
                ("step_base", C.c_int64), ("chain_offset", C.c_int64), ("seed", C.c_uint64),
                ("adapt_kind", C.c_int32), ("n_adapt", C.c_int32),
                ("adapt_target", C.c_float), ("adapt_rate", C.c_float),
                ("n_burnin", C.c_int32), ("thin", C.c_int32), ("n_samples", C.c_int32),
                ("trace_centered", C.c_int32), ("lanes_per_chain", C.c_int32), ("stats_batch", C.c_int32),
                ("trace_chains", C.c_int32), ("reserved", C.c_int32)]


class HmcIO(C.Structure):
    _fields_ = [("q", C.c_void_p), ("grad", C.c_void_p), ("logp", C.c_void_p), ("adapt", C.c_void_p),
                ("rng", C.c_void_p), ("accept_count", C.c_void_p), ("eps0", C.c_void_p),
                ("trace", C.c_void_p), ("trace_accept", C.c_void_p), ("stats", C.c_void_p),
                ("rec_accept_count", C.c_void_p)]


class InterleavedIO(C.Structure):
    _fields_ = [("k0", HmcIO), ("adapt1", C.c_void_p), ("accept_count1", C.c_void_p),
                ("eps0_1", C.c_void_p), ("trace_accept1", C.c_void_p), ("rec_accept_count1", C.c_void_p)]


class ViConfig(C.Structure):
    _fields_ = [("n_lr", C.c_int32), ("n_steps", C.c_int32), ("n_mc", C.c_int32),
                ("learn_a", C.c_int32), ("tied_b", C.c_int32), ("a_prior", C.c_int32),
                ("seed", C.c_uint64)]


class ViIO(C.Structure):
    _fields_ = [("lr", C.c_void_p), ("loc", C.c_void_p), ("rho", C.c_void_p), ("w", C.c_void_p),
                ("wb", C.c_void_p), ("elbo", C.c_void_p), ("prior", C.c_void_p), ("a_group", C.c_void_p),
                ("b_group", C.c_void_p)]


# every symbol include/autoreparam.h declares
SYMBOLS = ["arp_version", "arp_last_error", "arp_model_create", "arp_model_destroy", "arp_model_dim",
           "arp_model_logp_const", "arp_model_set_param", "arp_model_set_option", "arp_logp_grad", "arp_transform",
           "arp_hmc_run", "arp_interleaved_run", "arp_model_check", "arp_vi_run", "arp_vi_geometry", "arp_vi_attempts", "arp_relay_geometry", "arp_ess", "arp_ess_ws", "arp_ess_workspace_bytes",
           "arp_adapt_probe", "arp_clock_probe"]

_lib = None


def lib():
    """Load (once) and return the HIP engine; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "autoreparam_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.arp_version.restype = C.c_int
    L.arp_last_error.restype = C.c_char_p
    L.arp_model_create.argtypes = [C.POINTER(Dataset), C.POINTER(C.c_void_p)]
    L.arp_model_destroy.argtypes = [C.c_void_p]
    L.arp_model_dim.argtypes = [C.c_void_p]
    L.arp_model_logp_const.argtypes = [C.c_void_p, C.c_int]
    L.arp_model_logp_const.restype = C.c_double
    L.arp_model_set_param.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
    L.arp_model_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.arp_logp_grad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_void_p]
    L.arp_transform.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.arp_hmc_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(HmcConfig), C.POINTER(HmcIO), C.c_void_p]
    L.arp_interleaved_run.argtypes = [C.c_void_p, C.POINTER(HmcConfig), C.c_int,
                                      C.POINTER(InterleavedIO), C.c_void_p]
    L.arp_vi_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(ViConfig), C.POINTER(ViIO), C.c_void_p]
    L.arp_vi_geometry.argtypes = [C.POINTER(C.c_int32)]
    L.arp_vi_attempts.argtypes = [C.POINTER(C.c_int32)]
    L.arp_relay_geometry.argtypes = [C.POINTER(C.c_int32)]
    L.arp_model_check.argtypes = [C.c_void_p]
    L.arp_ess.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.arp_ess_ws.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.arp_ess_ws.restype = C.c_int
    L.arp_ess_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
    L.arp_ess_workspace_bytes.restype = C.c_int64
    L.arp_clock_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.arp_adapt_probe.argtypes = [C.POINTER(HmcConfig), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError("autoreparam engine: " + lib().arp_last_error().decode("utf-8", "replace"))
