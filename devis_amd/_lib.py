"""ctypes binding of libdevis_hip.so (include/devis_hip.h).

The product path has no CPU fallback: if the library is missing, or a call
returns a negative status, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEVIS_HIP_LIB: another build of the same library (A/B timing of kernel variants, tools/build_alt.sh)
LIB_PATH = os.environ.get("DEVIS_HIP_LIB") or os.path.join(_HERE, "libdevis_hip.so")

DEVIS_F32, DEVIS_BF16 = 0, 1
LABEL_I64, LABEL_U8, LABEL_I32 = 0, 1, 2
EVID_MAX_K = 8
EVID_SCAL_FLOATS = 4 + 3 * EVID_MAX_K

_STATUS = {-1: "bad argument", -2: "unsupported shape", -3: "HIP error", -4: "workspace too small",
           -5: "input exceeds the fixed scratch of the no-allocation entry point (N <= 1024 samples)"}


class DevisError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("N", C.c_int64),
                ("ID", C.c_int64), ("IH", C.c_int64), ("IW", C.c_int64),
                ("OD", C.c_int64), ("OH", C.c_int64), ("OW", C.c_int64),
                ("C1", C.c_int64), ("C2", C.c_int64), ("Cout", C.c_int64),
                ("KD", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
                ("SD", C.c_int32), ("SH", C.c_int32), ("SW", C.c_int32),
                ("PD", C.c_int32), ("PH", C.c_int32), ("PW", C.c_int32),
                ("dtype", C.c_int32)]


class ConvTDesc(C.Structure):
    _fields_ = [("N", C.c_int64), ("ID", C.c_int64), ("IH", C.c_int64), ("IW", C.c_int64),
                ("Cin", C.c_int64), ("Cout", C.c_int64),
                ("KD", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("dtype", C.c_int32)]


class ConvStats(C.Structure):
    _fields_ = [("mode", C.c_int32), ("reserved", C.c_int32), ("partials", C.c_void_p), ("partials_bytes", C.c_size_t),
                ("chunks", C.c_int64), ("x_ref", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p)]


class EvidParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("input_kind", C.c_int), ("disentangle", C.c_int),
                ("lam", C.c_float), ("a_au", C.c_float), ("eps", C.c_float)]


CRIT_DCE, CRIT_DCEU, CRIT_CE, CRIT_MSE, CRIT_KL = range(5)
PACK_CONV, PACK_CONV_DGRAD, PACK_CONVT, PACK_CONVT_DGRAD = range(4)
PACK_CHUNK = 1024
PACK_ROW_FLAG, PACK_ROW_MAX = 0x40000000, 8192

_P, _I64, _I, _F, _SZ = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t
_CD, _CTD, _EP = C.POINTER(ConvDesc), C.POINTER(ConvTDesc), C.POINTER(EvidParams)
_CS = C.POINTER(ConvStats)

# name -> (restype, argtypes); every symbol include/devis_hip.h declares
SIGNATURES = {
    "devis_version": (_I, []),
    "devis_last_hip_error": (_I, []),
    "devis_last_hip_error_string": (C.c_char_p, []),
    "devis_ncx_to_nxc": (_I, [_P, _P, _I64, _I64, _I64, _I, _P]),
    "devis_nxc_to_ncx": (_I, [_P, _P, _I64, _I64, _I64, _I, _P]),
    "devis_pack_conv_weight": (_I, [_P, _P, _I64, _I64, _I64, _I, _P]),
    "devis_pack_convT_weight": (_I, [_P, _P, _I64, _I64, _I64, _I, _P]),
    "devis_pack_weights_batch": (_I, [_P, _P, _I64, _I, _P]),
    "devis_conv_fwd": (_I, [_CD, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "devis_conv_dgrad": (_I, [_CD, _P, _P, _P, _P, _P, _P, _P]),
    "devis_conv_stats_chunks": (_I64, [_CD, _I, _I, _I, _I]),
    "devis_conv_fwd_stats": (_I, [_CD, _P, _P, _P, _P, _P, _P, _P, _P, _CS, _P]),
    "devis_conv_dgrad_stats": (_I, [_CD, _P, _P, _P, _P, _P, _P, _CS, _P]),
    "devis_conv_dgrad_normbwd": (_I, [_CD, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _CS, _P]),
    "devis_repack_conv_weight_dgrad": (_I, [_P, _P, _I64, _I64, _I64, _I, _P]),
    "devis_conv_wgrad_workspace_bytes": (_SZ, [_CD]),
    "devis_conv_wgrad": (_I, [_CD, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "devis_conv_wgrad_stage1": (_I, [_CD, _P, _P, _P, _P, _I, _P, _SZ, _P]),
    "devis_convT_fwd": (_I, [_CTD, _P, _P, _P, _P, _P]),
    "devis_convT_dgrad": (_I, [_CTD, _P, _P, _P, _P, _P]),
    "devis_convT_wgrad_workspace_bytes": (_SZ, [_CTD]),
    "devis_convT_wgrad": (_I, [_CTD, _P, _P, _P, _P, _P, _SZ, _P]),
    "devis_upcat_supported": (_I, [_I64, _I64, _I64, _I64, _I64, _I]),
    "devis_upcat_prepare": (_I, [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P]),
    "devis_upcat_fwd": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I, _P]),
    "devis_upcat_bwd_data": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I, _P]),
    "devis_upcat_grad_combine": (_I, [_P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P]),
    "devis_moments_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "devis_channel_moments": (_I, [_P, _I64, _I64, _I64, _I, _P, _P, _SZ, _P]),
    "devis_gn_finalize": (_I, [_P, _I64, _I64, _I64, _I64, _P, _P, _F, _P, _P, _P, _P, _P]),
    "devis_bn_finalize": (_I, [_P, _I64, _I64, _I64, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P]),
    "devis_affine_act": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _P]),
    "devis_norm_bwd_moments": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _P, _SZ, _P]),
    "devis_norm_bwd_finalize": (_I, [_P, _I64, _I64, _I64, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "devis_channel_moments_partials": (_I, [_P, _I64, _I64, _I64, _I, _P, _SZ, C.POINTER(C.c_int64), _P]),
    "devis_norm_bwd_moments_partials": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _SZ, C.POINTER(C.c_int64), _P]),
    "devis_gn_finalize_partials": (_I, [_P, _I64, _I64, _I64, _I64, _I64, _P, _P, _F, _P, _P, _P, _P, _P]),
    "devis_bn_finalize_partials": (_I, [_P, _I64, _I64, _I64, _I64, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "devis_norm_bwd_finalize_partials": (_I, [_P, _I64, _I64, _I64, _I64, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "devis_norm_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P]),
    "devis_add": (_I, [_P, _P, _P, _I64, _I, _P]),
    "devis_mul": (_I, [_P, _P, _P, _I64, _I, _P]),
    "devis_maxpool2d_fwd": (_I, [_P, _P, _P, _I64, _I64, _I64, _I64, _I, _P]),
    "devis_maxpool2d_bwd": (_I, [_P, _P, _P, _I64, _I64, _I64, _I64, _I, _P]),
    "devis_maxpool2d_bwd_add": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P]),
    "devis_evid_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "devis_evid_loss_fwd": (_I, [_P, _I64, _I64, _I64, _I, _P, _I, _I64, _I64, _I64, _F, _P, _P, _P, _P, _SZ, _P]),
    "devis_evid_loss_fwd_noalloc": (_I, [_P, _I64, _I64, _I64, _I, _P, _I, _I64, _I64, _I64, _F, _P, _P, _P, _P]),
    "devis_evid_loss_bwd": (_I, [_P, _I64, _I64, _I64, _I, _P, _I, _I64, _I64, _I64, _F, _P, _P, _P, _P, _P, _P, _P]),
    "devis_evid_crit_fwd": (_I, [_EP, _P, _I64, _I64, _I64, _I, _P, _P, _I, _I64, _I64, _I64, _P, _P, _P, _P, _SZ, _P]),
    "devis_evid_crit_bwd": (_I, [_EP, _P, _I64, _I64, _I64, _I, _P, _P, _I, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "devis_evid_infer": (_I, [_P, _I64, _I64, _I64, _I, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _I, _P]),
    "devis_att_chunks": (_I64, [_I64]),
    "devis_att_psi_bwd_chunks": (_I64, [_I64, _I64, _I]),
    "devis_att_psi_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _P, _SZ, _P]),
    "devis_att_gate_fwd": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P]),
    "devis_att_gate_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I64, _I64, _I64, _I, _P]),
    "devis_att_psi_bwd_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I]),
    "devis_att_psi_bwd": (_I, [_P] * 12 + [_I64, _I64, _I64, _I, _P, _P, _P, _SZ, _P, _P, _P, _SZ, _P]),
    "devis_eval_counts_workspace_bytes": (_SZ, [_I64]),
    "devis_eval_counts": (_I, [_P, _P, _I, _P, _I, _P, _I64, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "devis_adam_step": (_I, [_P, _P, _P, _P, _I64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _I64, _P]),
    "devis_softplus_bwd": (_I, [_P, _P, _P, _I64, _P]),
}

_lib = None


# ---- launch recording (devis_amd/launch_plan.py): while a thread records, load() hands it a proxy that notes every
# launching call (function, arguments) after making it.  Host-only queries are not noted.
_rec_tls = threading.local()         # .rec: the thread's Recorder
_NOT_LAUNCHES = ("_bytes", "_chunks", "_supported")


class _RecordingLib:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name.endswith(_NOT_LAUNCHES) or name in ("devis_version", "devis_last_hip_error", "devis_last_hip_error_string"):
            call = fn
        else:
            def call(*args, _fn=fn):
                rc = _fn(*args)
                rec = getattr(_rec_tls, "rec", None)
                if rec is not None and not rec.opaque:
                    rec.ops.append((_fn, list(args), True))
                return rc
        setattr(self, name, call)
        return call


_recording_lib = None


def load():
    """Load the HIP library; raise (never fall back) if it is not there."""
    global _lib, _recording_lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DevisError(
                "libdevis_hip.so not found at %s - build it with devis_amd/csrc/build.sh "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)    # AttributeError if a declared symbol is missing
            except AttributeError:
                if not os.environ.get("DEVIS_HIP_LIB"):
                    raise
                continue                   # an older build loaded on purpose for A/B timing: calls to it raise later
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    if getattr(_rec_tls, "rec", None) is not None:     # thread-local only: no shared counter to lose an update of
        if _recording_lib is None:
            _recording_lib = _RecordingLib(_lib)
        return _recording_lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = _STATUS.get(rc, "status %d" % rc)
        if rc == -3:
            msg += ": " + lib.devis_last_hip_error_string().decode()
        raise DevisError("%s failed: %s" % (what, msg))
