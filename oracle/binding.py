"""ctypes binding of oracle/libkcforacle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Result(C.Structure):
    _fields_ = [("total_kmers", C.c_int32), ("eff_length", C.c_int32), ("obs", C.c_int32), ("var", C.c_int32),
                ("inner", C.c_int32), ("left", C.c_int32), ("right", C.c_int32), ("_pad", C.c_int32),
                ("kmer_count_sum", C.c_int64)]


RESULT_DTYPE = np.dtype([("total_kmers", "<i4"), ("eff_length", "<i4"), ("obs", "<i4"), ("var", "<i4"),
                         ("inner", "<i4"), ("left", "<i4"), ("right", "<i4"), ("_pad", "<i4"),
                         ("kmer_count_sum", "<i8")])
INT_FIELDS = ("total_kmers", "eff_length", "obs", "var", "inner", "left", "right", "kmer_count_sum")


class Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kmer_length", "mode", "counter_size", "lut_prefix_length", "signature_length",
                                          "min_count", "max_count", "both_strands", "version", "n_bins")] + \
               [("total_kmers", C.c_int64), ("lut_entries", C.c_int64)]


def build(force: bool = False) -> str:
    if os.environ.get("KCF_ORACLE_SO"):                 # the sanitizer build (tests/test_sanitizers.py)
        return os.environ["KCF_ORACLE_SO"]
    so = os.path.join(_HERE, "libkcforacle.so")
    src = [os.path.join(_HERE, f) for f in ("kcf_oracle.c", "kcf_oracle.h")]
    def stale():
        return not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale():
        # one builder at a time: the ranks of a multi-process test import this module side by side, and a fresh checkout gives
        # sources and library timestamps in any order -- a second `make` must not overwrite the file a sibling is loading
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                subprocess.run(["make", "-C", _HERE, "libkcforacle.so"] + (["-B"] if force else []), check=True, capture_output=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.kcfo_last_error.restype = C.c_char_p
        L.kcfo_db_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.kcfo_db_close.argtypes = [C.c_void_p]
        L.kcfo_db_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.kcfo_get_count.argtypes = [C.c_void_p, C.c_uint64]
        L.kcfo_get_count.restype = C.c_int32
        L.kcfo_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kcfo_dump.restype = C.c_int64
        L.kcfo_norm_table.argtypes = [C.c_int, C.c_void_p]
        L.kcfo_kmer_encode.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]
        L.kcfo_kmer_revcomp.argtypes = [C.c_uint64, C.c_int]
        L.kcfo_kmer_revcomp.restype = C.c_uint64
        L.kcfo_kmer_canonical.argtypes = [C.c_uint64, C.c_int]
        L.kcfo_kmer_canonical.restype = C.c_uint64
        L.kcfo_kmer_signature.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        L.kcfo_kmer_signature.restype = C.c_int32
        L.kcfo_kmer_prefix.argtypes = [C.c_uint64, C.c_int]
        L.kcfo_kmer_prefix.restype = C.c_int32
        L.kcfo_kmer_suffix.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        L.kcfo_kmers_list.argtypes = [C.c_char_p, C.c_int32, C.c_int, C.c_void_p]
        L.kcfo_kmers_list.restype = C.c_int32
        L.kcfo_eff_length.argtypes = [C.c_char_p, C.c_int32, C.c_int]
        L.kcfo_eff_length.restype = C.c_int32
        L.kcfo_distance.argtypes = [C.c_int, C.c_int32]
        L.kcfo_distance.restype = C.c_int32
        L.kcfo_process_window.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(Result)]
        L.kcfo_accumulate.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int32, C.POINTER(Result)]
        L.kcfo_score.argtypes = [C.POINTER(Result), C.c_double, C.c_double, C.c_double,
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.kcfo_windows.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.kcfo_windows.restype = C.c_int64
        L.kcfo_screen.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        _LIB = L
    return _LIB


class OracleDb:
    """KMC.java as restated by kcf_oracle.c."""

    def __init__(self, prefix: str):
        self._h = C.c_void_p()
        rc = lib().kcfo_db_open(prefix.encode(), C.byref(self._h))
        if rc:
            raise RuntimeError(lib().kcfo_last_error().decode())
        self.info = Info()
        lib().kcfo_db_info(self._h, C.byref(self.info))

    def close(self):
        if self._h:
            lib().kcfo_db_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()

    def get_count(self, kmer: int) -> int:
        return lib().kcfo_get_count(self._h, C.c_uint64(int(kmer)))

    def dump(self):
        n = self.info.total_kmers
        km = np.zeros(n, dtype=np.uint64)
        ct = np.zeros(n, dtype=np.int32)
        got = lib().kcfo_dump(self._h, km.ctypes.data, ct.ctypes.data)
        return km[:got], ct[:got]

    def process_window(self, text: bytes, min_count: int = 1) -> dict:
        r = Result()
        rc = lib().kcfo_process_window(self._h, text, len(text), min_count, C.byref(r))
        if rc:
            raise RuntimeError(lib().kcfo_last_error().decode())
        return {f: getattr(r, f) for f in INT_FIELDS}

    def screen(self, text: bytes, win_begin, win_len, min_count: int = 1, threads: int = 1) -> np.ndarray:
        wb = np.ascontiguousarray(win_begin, dtype=np.int64)
        wl = np.ascontiguousarray(win_len, dtype=np.int32)
        out = np.zeros(len(wb), dtype=RESULT_DTYPE)
        rc = lib().kcfo_screen(self._h, text, wb.ctypes.data, wl.ctypes.data, len(wb), min_count, threads, out.ctypes.data)
        if rc:
            raise RuntimeError(lib().kcfo_last_error().decode())
        return out


def accumulate(counts, k: int, min_count: int = 1) -> dict:
    c = np.ascontiguousarray(counts, dtype=np.int32)
    r = Result()
    lib().kcfo_accumulate(c.ctypes.data, len(c), k, min_count, C.byref(r))
    return {f: getattr(r, f) for f in INT_FIELDS}


def score(res: dict, wi=0.3, wt=0.3, wr=0.4):
    r = Result()
    for f in INT_FIELDS:
        setattr(r, f, int(res[f]))
    s, m = C.c_double(), C.c_double()
    bad = lib().kcfo_score(C.byref(r), wi, wt, wr, C.byref(s), C.byref(m))
    return s.value, m.value, bad


def score_array(res: np.ndarray, wi=0.3, wt=0.3, wr=0.4):
    sc = np.zeros(len(res))
    md = np.zeros(len(res))
    for i in range(len(res)):
        r = Result()
        for f in INT_FIELDS:
            setattr(r, f, int(res[f][i]))
        s, m = C.c_double(), C.c_double()
        lib().kcfo_score(C.byref(r), wi, wt, wr, C.byref(s), C.byref(m))
        sc[i], md[i] = s.value, m.value
    return sc, md


def norm_table(m: int) -> np.ndarray:
    out = np.zeros(4 ** m, dtype=np.int32)
    lib().kcfo_norm_table(m, out.ctypes.data)
    return out


def windows(seq_len: int, window: int, step: int, k: int):
    n = lib().kcfo_windows(seq_len, window, step, k, None, None, 0)
    if n < 0:
        raise ValueError("window <= k-1: the reference never terminates")
    s = np.zeros(n, dtype=np.int32)
    e = np.zeros(n, dtype=np.int32)
    lib().kcfo_windows(seq_len, window, step, k, s.ctypes.data, e.ctypes.data, n)
    return s, e
