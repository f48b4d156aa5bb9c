"""ctypes binding of the C ABI declared in include/clairvoyante_amd.h.

The library is required: there is deliberately no fallback implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CV_HIP_LIB: another build of the SAME library (development: two builds timed on one GPU box, tools/gpu_lib_ab.sh)
_DEFAULT_LIB = os.path.join(_HERE, "csrc", "libclairvoyante_hip.so")
LIB_PATH = os.environ.get("CV_HIP_LIB") or _DEFAULT_LIB

NUM_PARAMS = 18
NUM_OUT = 16


class CvArch(ctypes.Structure):
    _fields_ = [("kh", ctypes.c_int32 * 3), ("cout", ctypes.c_int32 * 3), ("pool", ctypes.c_int32 * 3),
                ("fc4", ctypes.c_int32), ("fc5", ctypes.c_int32)]


class CvError(RuntimeError):
    pass


_lib = None

_SIGS = {
    "cv_last_error": (ctypes.c_char_p, []),
    "cv_create": (ctypes.c_int, [ctypes.POINTER(CvArch), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "cv_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "cv_param_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)]),
    "cv_param_buffer": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "cv_set_param": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_void_p]),
    "cv_get_param": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_void_p]),
    "cv_params_changed": (ctypes.c_int, [ctypes.c_void_p]),
    "cv_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                  ctypes.c_void_p]),
    "cv_call_postproc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "cv_get_activation": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_void_p]),
    "cv_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    "cv_get_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "cv_kernel_times": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_int64)]),
    "cv_loss": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                               ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]),
    "cv_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                               ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64,
                               ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]),
    "cv_grad_buffer": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_int64)]),
    "cv_grad_bucket_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "cv_bind_grad_bucket": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "cv_grad_async": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_void_p, ctypes.c_void_p]),
    "cv_loss_accumulate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "cv_loss_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_void_p]),
    "cv_kernel_name": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]),
    "cv_selu_sweep": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
                                     ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "cv_apply_adam": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_int64,
                                     ctypes.c_void_p]),
    "cv_apply_adam_accumulate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_int64,
                                     ctypes.c_void_p]),
    "cv_flat_copy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_void_p]),
    "cv_adam_buffers": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]),
    "cv_parse_tensor_text": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "cv_set_host_threads": (ctypes.c_int, [ctypes.c_int]),
    "cv_blosc_nbytes": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64]),
    "cv_blosc_decompress": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]),
    "cv_blosc_decompress_many": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int64, ctypes.c_void_p]),
    "cv_blosc_unpack_blocks": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "cv_blosc_compress_lz4": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    "cv_crc32c": (ctypes.c_uint32, [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int64]),
    "cv_format_vcf": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64),
                                     ctypes.POINTER(ctypes.c_int64)]),
    # pileup front end
    "cv_pileup_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_void_p)]),
    "cv_pileup_destroy": (None, [ctypes.c_void_p]),
    "cv_pileup_set_reference": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64]),
    "cv_pileup_set_candidates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "cv_pileup_add_sam": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "cv_pileup_pending": (ctypes.c_int64, [ctypes.c_void_p]),
    "cv_pileup_flush": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "cv_pileup_finish": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_void_p]),
    "cv_pileup_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)]),
    "cv_pileup_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    "cv_pileup_set_contig": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "cv_pileup_extract_candidates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                                    ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "cv_pileup_get_extracted": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]),
    "cv_pileup_adopt_candidates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                  ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "cv_pileup_recount": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "cv_pileup_get_candidates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.POINTER(ctypes.c_int64)]),
    "cv_bam_open": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "cv_bam_close": (None, [ctypes.c_void_p]),
    "cv_bam_nref": (ctypes.c_int, [ctypes.c_void_p]),
    "cv_bam_ref": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                  ctypes.POINTER(ctypes.c_int64)]),
    "cv_bam_has_index": (ctypes.c_int, [ctypes.c_void_p]),
    "cv_bam_view_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_int]),
    "cv_bam_view_read": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]),
    "cv_bam_view_records": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]),
    "cv_bam_record_cigar": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]),
    "cv_inflate_raw": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]),
    "cv_inflate_stream": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32)]),
    "cv_crc32_ieee": (ctypes.c_uint32, [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int64]),
    "cv_pileup_add_bam": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int64)]),
    "cv_format_tensor_row": (ctypes.c_int64, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64,
                                              ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
}

EXPORTS = sorted(_SIGS)


def load():
    """dlopen the HIP library (built by clairvoyante_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CvError("%s is missing: build it with `python -m clairvoyante_amd.build` "
                      "(hipcc, gfx950); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64; it must be the one HIP runtime of the process
    # (two runtimes = two device contexts and "no ROCm-capable device"), so torch is imported --
    # and its runtime mapped -- before this library, whose libamdhip64 dependency then
    # resolves to the copy already loaded.
    import torch  # noqa: F401
    if os.path.abspath(LIB_PATH) != os.path.abspath(_DEFAULT_LIB):
        import logging
        logging.warning("clairvoyante_amd: CV_HIP_LIB overrides the HIP library: loading %s instead of %s",
                        LIB_PATH, _DEFAULT_LIB)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError = a declared entry point is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    lib.cv_set_host_threads(min(usable_cores(), 16))
    return lib


def usable_cores():
    """host threads this process may actually use: affinity mask capped by the cgroup CPU quota (a box can
    report 256 logical CPUs under a 16-CPU quota; oversubscribing it throttles)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(float(quota) / float(period)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / per))))
        except Exception:
            pass
    return n


def check(rc):
    if rc != 0:
        msg = load().cv_last_error()
        raise CvError(msg.decode("utf-8", "replace") if msg else "clairvoyante_amd call failed")
