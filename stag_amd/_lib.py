"""ctypes binding of libstag_hip.so (include/stag_hip.h).  There is no fallback: if the
library is missing or no gfx950 GPU is usable, calls raise."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# STAG_HIP_LIB selects another build of the same library (kernel ablation variants for profiling)
LIB_PATH = os.environ.get("STAG_HIP_LIB") or os.path.join(_HERE, "lib", "libstag_hip.so")

STAG_OK = 0
SEQ_OK, SEQ_MISSING_CLASSIFIER, SEQ_STRING_CLASSIFIER, SEQ_RANKCUT_SHAPE, SEQ_EMPTY_PATH, \
    SEQ_ZERO_COLUMNS, SEQ_FILTERED, SEQ_LENGTH_MISMATCH, SEQ_BACKTRANSLATE = range(9)
PROF_CLASSES = ("encode", "walk_level", "group", "rankcut", "other")

c_p = ctypes.c_void_p


class StagError(RuntimeError):
    pass


class Tables(ctypes.Structure):
    _fields_ = [("n_nodes", ctypes.c_int32), ("n_features", ctypes.c_int32),
                ("child_first", c_p), ("row_kind", c_p), ("intercept", c_p),
                ("row_off", c_p), ("row_feat", c_p), ("row_w", c_p),
                ("n_keys", ctypes.c_int32), ("n_tf", ctypes.c_int32), ("tf", c_p)]


class Results(ctypes.Structure):
    _fields_ = [("path_node", c_p), ("path_prob", c_p), ("path_len", c_p), ("n_lead_single", c_p),
                ("level_prob", c_p), ("selected", c_p), ("n_aligned", c_p), ("status", c_p)]


EXPORTS = {
    # name: (restype, argtypes)
    "stag_last_error": (ctypes.c_char_p, []),
    "stag_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int32)]),
    "stag_db_create": (ctypes.c_int, [ctypes.POINTER(Tables), ctypes.c_int32, ctypes.POINTER(c_p)]),
    "stag_db_destroy": (None, [c_p]),
    "stag_db_info": (ctypes.c_int, [c_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                    ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]),
    "stag_db_reserve": (ctypes.c_int, [c_p, ctypes.c_int64, ctypes.c_int64]),
    "stag_words_per_row": (ctypes.c_int32, [ctypes.c_int32]),
    "stag_encode_a2m": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, c_p, c_p, ctypes.c_int64,
                                       ctypes.c_double, c_p, c_p, c_p, c_p]),
    "stag_encode_a2m_dev": (ctypes.c_int, [ctypes.c_int32, c_p, ctypes.c_int64, c_p, ctypes.c_int64,
                                           ctypes.c_double, c_p, c_p, c_p, c_p, c_p]),
    "stag_classify_bits": (ctypes.c_int, [c_p, c_p, ctypes.c_int64, c_p, ctypes.POINTER(Results)]),
    "stag_classify_bits_dev": (ctypes.c_int, [c_p, c_p, ctypes.c_int64, c_p, ctypes.POINTER(Results), c_p]),
    "stag_parse_onehot_dev": (ctypes.c_int, [ctypes.c_int32, c_p, ctypes.c_int64, c_p, c_p, ctypes.c_int64, c_p, c_p, c_p, c_p]),
    "stag_onehot_text_parse": (ctypes.c_int, [c_p, c_p, ctypes.c_int64, c_p, c_p, ctypes.c_int64, c_p, c_p]),
    "stag_onehot_text_classify": (ctypes.c_int, [c_p, ctypes.c_int64, c_p, c_p, ctypes.c_int64, c_p, ctypes.POINTER(Results)]),
    "stag_classify_a2m": (ctypes.c_int, [c_p, c_p, c_p, ctypes.c_int64, ctypes.c_double,
                                         ctypes.POINTER(Results), c_p, c_p]),
    "stag_classify_a2m_dev": (ctypes.c_int, [c_p, c_p, ctypes.c_int64, c_p, ctypes.c_int64, ctypes.c_double,
                                             ctypes.POINTER(Results), c_p]),
    "stag_classify_a2m_trimmed": (ctypes.c_int, [c_p, c_p, c_p, c_p, ctypes.c_int64, ctypes.c_double,
                                                 ctypes.POINTER(Results), c_p, c_p]),
    "stag_classify_a2m_trimmed_dev": (ctypes.c_int, [c_p, c_p, ctypes.c_int64, c_p, c_p, ctypes.c_int64, ctypes.c_double,
                                                     ctypes.POINTER(Results), c_p]),
    "stag_classify_a2m_noalloc_dev": (ctypes.c_int, [c_p, c_p, ctypes.c_int64, c_p, ctypes.c_int64, ctypes.c_double,
                                                     ctypes.POINTER(Results), c_p]),
    "stag_backtranslate": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64,
                                          ctypes.c_int32, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "stag_backtranslate_dev": (ctypes.c_int, [c_p, ctypes.c_int64, c_p, c_p, c_p, ctypes.c_int64, ctypes.c_int32, c_p,
                                              ctypes.c_int64, c_p, c_p, c_p, c_p]),
    "stag_classify_protein_a2m": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, ctypes.c_int64, ctypes.c_double,
                                                 ctypes.POINTER(Results), c_p, c_p]),
    "stag_classify_protein_a2m_dev": (ctypes.c_int, [c_p, c_p, ctypes.c_int64, c_p, c_p, c_p, ctypes.c_int64,
                                                     ctypes.c_double, ctypes.POINTER(Results), c_p]),
    "stag_classify_concat": (ctypes.c_int, [c_p, ctypes.c_int32, c_p, c_p, c_p, c_p, c_p, ctypes.c_int64,
                                            ctypes.POINTER(Results), c_p]),
    "stag_host_alloc": (c_p, [ctypes.c_int64]),
    "stag_host_free": (None, [c_p]),
    "stag_profile_enable": (ctypes.c_int, [c_p, ctypes.c_int32]),
    "stag_profile_read": (ctypes.c_int, [c_p, c_p, c_p]),
}

_lib = None


def lib():
    """Load libstag_hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C stag_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StagError("%s is missing: build it with `make -C stag_amd/csrc` "
                            "(there is no CPU fallback for this path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != STAG_OK:
        msg = lib().stag_last_error()
        raise StagError("libstag_hip error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else "?"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(c_p)


def kernel_source_sha1():
    """Hash of the sources libstag_hip.so is built from (they travel with the library): what a committed
    profile is stamped with, so that numbers measured on other kernels are not mistaken for this build's."""
    import hashlib
    h = hashlib.sha1()
    for name in ("stag_kernels.hip.h", "stag_hip.hip"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def device_count():
    n = ctypes.c_int32(0)
    lib().stag_device_count(ctypes.byref(n))
    return n.value


class _PinnedBlock:
    """One stag_host_alloc allocation; freed when the last array over it is gone."""

    def __init__(self, nbytes):
        self.p = lib().stag_host_alloc(nbytes)
        if not self.p:
            raise StagError("stag_host_alloc(%d) failed: %s" % (nbytes, lib().stag_last_error().decode()))

    def __del__(self):
        try:
            if self.p:
                lib().stag_host_free(self.p)
                self.p = None
        except Exception:
            pass


class PinnedArray:
    """A NumPy array over page-locked host memory from stag_host_alloc: buffers of this kind let the host
    entry points copy by DMA beside the kernels.  .a is the array.  The memory belongs to the array and its
    views (array -> ctypes buffer -> block): it is freed when the last of them is gone, not when this object is."""

    def __init__(self, shape, dtype):
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        block = _PinnedBlock(max(self.nbytes, 16))
        buf = (ctypes.c_uint8 * max(self.nbytes, 16)).from_address(block.p)
        buf._stag_block = block          # every view of the array keeps `buf` alive, `buf` keeps the block
        self.a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
