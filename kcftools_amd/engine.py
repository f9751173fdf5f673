"""ctypes binding of libkcfx.so (include/kcfx.h) -- the only way Python reaches the engine.

There is no Python or CPU implementation of the hot path here: if the HIP library is missing or no
GPU is usable, construction of `KmcDb` / `Reference` raises.  The class / method names follow the
reference's objects for this path: `KmcDb` <-> Data/KMC.java, `Reference` <-> the window list +
FastaIndex-backed text of Plugins/GetVariants.java, `screen` <-> the processWindow fan-out.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KCFX_LIB") or os.path.join(_HERE, "libkcfx.so")   # KCFX_LIB: experiments build only

RESULT_DTYPE = np.dtype([("total_kmers", "<i4"), ("eff_length", "<i4"), ("obs", "<i4"), ("var", "<i4"),
                         ("inner", "<i4"), ("left", "<i4"), ("right", "<i4"), ("_pad", "<i4"),
                         ("kmer_count_sum", "<i8")])
INT_FIELDS = ("total_kmers", "eff_length", "obs", "var", "inner", "left", "right", "kmer_count_sum")
assert RESULT_DTYPE.itemsize == 40

# every symbol include/kcfx.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "kcfx_db_open", "kcfx_db_open_opts", "kcfx_db_build", "kcfx_db_info", "kcfx_db_header", "kcfx_db_lookup", "kcfx_db_close",
    "kcfx_release_cached",
    "kcfx_ref_create", "kcfx_ref_create_opts", "kcfx_ref_n_windows", "kcfx_ref_n_kmer_slots", "kcfx_ref_destroy",
    "kcfx_screen_ref", "kcfx_screen_ref_async", "kcfx_screen_ref_multi", "kcfx_screen_ref_multi_async", "kcfx_screen",
    "kcfx_ref_reserve_multi",
    "kcfx_cohort_open", "kcfx_cohort_info", "kcfx_cohort_close", "kcfx_cohort_screen_ref", "kcfx_cohort_screen_ref_async",
    "kcfx_ctx_create", "kcfx_ctx_n_devices", "kcfx_ctx_destroy", "kcfx_screen_sharded", "kcfx_shard_plan",
    "kcfx_db_open_slice", "kcfx_table_slice_of", "kcfx_screen_ref_partial_async", "kcfx_screen_ref_partial", "kcfx_ref_partial_bytes",
    "kcfx_ref_export_partial_async", "kcfx_ref_merge_partial_async", "kcfx_ref_merge_from", "kcfx_ref_finish_async", "kcfx_ref_finish",
    "kcfx_score", "kcfx_windows",
    "kcfx_table_plan", "kcfx_table_plan_for", "kcfx_set_cache_fraction", "kcfx_shard_slice_io", "kcfx_ref_import_partial_async", "kcfx_ref_tile_range", "kcfx_prof_enable", "kcfx_prof_read", "kcfx_ref_plan_async", "kcfx_ref_is_planned", "kcfx_prof_read_plan", "kcfx_last_error", "kcfx_device_count", "kcfx_version",
)


class DbInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kmer_length", "mode", "counter_size", "lut_prefix_length",
                                          "signature_length", "min_count", "max_count", "both_strands",
                                          "version", "n_bins")] + \
               [("total_kmers", C.c_int64), ("table_bytes", C.c_int64), ("table_lines", C.c_int64),
                ("slots_per_line", C.c_int32), ("device", C.c_int32), ("load_seconds", C.c_double),
                ("table_mode", C.c_int32), ("minimizer_length", C.c_int32),
                ("heavy_lines", C.c_int64), ("overflow_lines", C.c_int64),
                ("table_format", C.c_int32), ("lut_suspect", C.c_int32), ("stored_kmers", C.c_int64),
                ("stash_kmers", C.c_int64), ("stash_bytes", C.c_int64), ("multi_cell_kmers", C.c_int64),
                ("second_choice_kmers", C.c_int64), ("extended", C.c_int32), ("regions", C.c_int32),
                ("table_slices", C.c_int32), ("table_slice", C.c_int32)]


class ShardStats(C.Structure):
    _fields_ = [("shard_mode_used", C.c_int32), ("n_devices", C.c_int32), ("seconds_total", C.c_double),
                ("seconds_ref", C.c_double), ("seconds_db_open_sum", C.c_double), ("seconds_screen_sum", C.c_double),
                ("seconds_exposed_load", C.c_double)]


SHARD_AUTO, SHARD_SAMPLES, SHARD_WINDOWS, SHARD_GRID, SHARD_COHORT, SHARD_TABLE = 0, 1, 2, 3, 4, 5
SLICES_MAX = 64
COHORT_MAX = 8


class RefOpts(C.Structure):
    _fields_ = [("expected_screens", C.c_int32), ("reserved", C.c_int32)]


class DbOpts(C.Structure):
    """kcfx_db_opts"""
    _fields_ = [("expected_queries", C.c_int64), ("no_fallback", C.c_int32), ("reserved", C.c_int32)]


class CohortInfo(C.Structure):
    """kcfx_cohort_info_t"""
    _fields_ = [("n_samples", C.c_int32), ("kmer_length", C.c_int32), ("both_strands", C.c_int32), ("device", C.c_int32),
                ("total_records", C.c_int64), ("table_lines", C.c_int64), ("table_bytes", C.c_int64), ("stash_records", C.c_int64),
                ("load_seconds", C.c_double)]


class KcfxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"kcfx error {code}: {msg}")
        self.code = code


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7,
    the same SONAME libkcfx.so needs).  If libkcfx.so were loaded first it would bind /opt/rocm's copy, a later
    `import torch` would bring in the bundled copy as well, and the second runtime finds no GPU.  Pre-loading the
    bundled copy (when PyTorch is installed) makes both bind the same runtime, whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """Load libkcfx.so; fail loudly if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension is not built "
                              f"(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.kcfx_last_error.restype = C.c_char_p
        L.kcfx_version.restype = C.c_char_p
        L.kcfx_device_count.argtypes = [C.POINTER(i32)]
        L.kcfx_db_open.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
        L.kcfx_db_open_opts.argtypes = [C.c_char_p, i32, C.POINTER(DbOpts), C.POINTER(vp)]
        L.kcfx_db_open_slice.argtypes = [C.c_char_p, i32, C.POINTER(DbOpts), i32, i32, C.POINTER(vp)]
        L.kcfx_table_slice_of.argtypes = [C.c_uint32, i32, i32, i32]
        L.kcfx_screen_ref_partial_async.argtypes = [vp, vp, i32, vp]
        L.kcfx_screen_ref_partial.argtypes = [vp, vp, i32]
        L.kcfx_ref_partial_bytes.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
        L.kcfx_ref_partial_bytes.restype = i64
        L.kcfx_ref_export_partial_async.argtypes = [vp, vp, vp, vp]
        L.kcfx_ref_merge_partial_async.argtypes = [vp, vp, vp, vp]
        L.kcfx_ref_import_partial_async.argtypes = [vp, vp, vp, vp]
        L.kcfx_ref_tile_range.argtypes = [vp, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.kcfx_ref_merge_from.argtypes = [vp, vp]
        L.kcfx_ref_finish_async.argtypes = [vp, vp, vp]
        L.kcfx_ref_finish.argtypes = [vp, vp]
        L.kcfx_db_build.argtypes = [vp, vp, i64, i32, i32, i32, i32, C.POINTER(vp)]
        L.kcfx_db_info.argtypes = [vp, C.POINTER(DbInfo)]
        L.kcfx_db_header.argtypes = [C.c_char_p, C.POINTER(DbInfo)]
        L.kcfx_ctx_create.argtypes = [vp, i32, C.POINTER(vp)]
        L.kcfx_ctx_n_devices.argtypes = [vp]
        L.kcfx_ctx_destroy.argtypes = [vp]
        L.kcfx_ctx_destroy.restype = None
        L.kcfx_screen_sharded.argtypes = [vp, vp, i32, i32, vp, i64, vp, vp, i32, i32, i32, vp, C.POINTER(ShardStats)]
        L.kcfx_shard_plan.argtypes = [i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, i64, C.POINTER(i64), C.POINTER(i32)]
        L.kcfx_table_plan.argtypes = [i64, i32, i32, i64, C.POINTER(i64), C.POINTER(i32)]
        L.kcfx_table_plan_for.argtypes = [i64, i32, i32, i64, i64, C.POINTER(i64), C.POINTER(i32)]
        L.kcfx_set_cache_fraction.argtypes = [C.c_double]
        L.kcfx_set_cache_fraction.restype = C.c_double
        L.kcfx_shard_slice_io.argtypes = [C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.kcfx_db_lookup.argtypes = [vp, vp, i64, vp]
        L.kcfx_db_close.argtypes = [vp]
        L.kcfx_db_close.restype = None
        L.kcfx_release_cached.argtypes = [i32]
        L.kcfx_ref_create.argtypes = [i32, i32, vp, i64, vp, vp, i32, C.POINTER(vp)]
        L.kcfx_ref_create_opts.argtypes = [i32, i32, vp, i64, vp, vp, i32, C.POINTER(RefOpts), C.POINTER(vp)]
        L.kcfx_ref_n_windows.argtypes = [vp]
        L.kcfx_ref_n_kmer_slots.argtypes = [vp]
        L.kcfx_ref_n_kmer_slots.restype = i64
        L.kcfx_ref_destroy.argtypes = [vp]
        L.kcfx_ref_destroy.restype = None
        L.kcfx_screen_ref.argtypes = [vp, vp, i32, vp]
        L.kcfx_screen_ref_async.argtypes = [vp, vp, i32, vp, vp]
        L.kcfx_ref_reserve_multi.argtypes = [vp, i32]
        L.kcfx_screen_ref_multi.argtypes = [vp, i32, vp, i32, vp]
        L.kcfx_screen_ref_multi_async.argtypes = [vp, i32, vp, i32, vp, vp]
        L.kcfx_screen.argtypes = [vp, vp, i64, vp, vp, i32, i32, vp]
        L.kcfx_cohort_open.argtypes = [vp, i32, i32, C.POINTER(DbOpts), C.POINTER(vp)]
        L.kcfx_cohort_info.argtypes = [vp, C.POINTER(CohortInfo)]
        L.kcfx_cohort_close.argtypes = [vp]
        L.kcfx_cohort_close.restype = None
        L.kcfx_cohort_screen_ref.argtypes = [vp, vp, i32, vp]
        L.kcfx_cohort_screen_ref_async.argtypes = [vp, vp, i32, vp, vp]
        L.kcfx_score.argtypes = [vp, i64, dbl, dbl, dbl, vp, vp]
        L.kcfx_windows.argtypes = [i32, i32, i32, i32, vp, vp, i64, C.POINTER(i64)]
        L.kcfx_prof_enable.argtypes = [vp, i32]
        L.kcfx_prof_read.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), i32]
        L.kcfx_prof_read_plan.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), i32]
        L.kcfx_ref_plan_async.argtypes = [vp, vp]
        L.kcfx_ref_is_planned.argtypes = [vp]
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc != 0:
        raise KcfxError(rc, lib().kcfx_last_error().decode(errors="replace"))


def release_cached(device: int = -1) -> None:
    """Give the device buffers that closed databases / destroyed references left behind back to the driver."""
    _check(lib().kcfx_release_cached(device))


def table_plan(total_kmers: int, k: int, both_strands: bool = True, free_bytes: int = 256 << 30, expected_queries: int = 0) -> tuple[int, int]:
    """kcfx_table_plan_for (host only): (lines, regions) of the cell table kcfx_db_open_opts would build for a handle that serves
    expected_queries look-ups (0 = unknown: kcfx_db_open's table); lines = 0: a line table."""
    n, r = C.c_int64(0), C.c_int32(0)
    _check(lib().kcfx_table_plan_for(int(total_kmers), int(k), int(bool(both_strands)), int(free_bytes), int(expected_queries), C.byref(n), C.byref(r)))
    return n.value, r.value


def shard_slice_io() -> tuple[int, int, int]:
    """kcfx_shard_slice_io: (bytes of .kmc_suf read, chunks read, chunk requests served by a sibling loader's read) of the most recent
    table-sharded call of this process."""
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    _check(lib().kcfx_shard_slice_io(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def set_cache_fraction(fraction: float) -> float:
    """kcfx_set_cache_fraction: the share of a device that released buffers may occupy (default 0.5); returns the previous value."""
    return float(lib().kcfx_set_cache_fraction(float(fraction)))


def table_slice_of(g32: int, lbits: int, n_slices: int, slice_class: int = 0) -> int:
    """kcfx_table_slice_of (host only): which of n_slices devices holds the cells of a minimizer of slice class `slice_class`
    (0 .. 3: two hash bits of its order key) whose first-choice line and r are the 32 bits g32 in a table of 2^lbits lines per
    region; -1 on bad arguments."""
    return int(lib().kcfx_table_slice_of(C.c_uint32(g32 & 0xFFFFFFFF), int(lbits), int(n_slices), int(slice_class)))


def device_count() -> int:
    n = C.c_int32(0)
    _check(lib().kcfx_device_count(C.byref(n)))
    return n.value


def db_header(kmc_prefix: str) -> DbInfo:
    """Header fields of a KMC database (KMC.java:107-168) without touching a GPU."""
    info = DbInfo()
    _check(lib().kcfx_db_header(os.fsencode(kmc_prefix), C.byref(info)))
    return info


def shard_plan(win_len, k: int, n_samples: int, n_devices: int, mode: int = SHARD_AUTO):
    """kcfx_shard_plan (host only): list of (device index, sample, w_lo, w_hi) segments and the mode used."""
    wl = np.ascontiguousarray(win_len, dtype=np.int32)
    n, used = C.c_int64(0), C.c_int32(0)
    _check(lib().kcfx_shard_plan(mode, n_devices, n_samples, wl.ctypes.data if len(wl) else None, len(wl), k, None, None, None, None, 0,
                                 C.byref(n), C.byref(used)))
    arr = [np.zeros(max(n.value, 1), dtype=np.int32) for _ in range(4)]
    _check(lib().kcfx_shard_plan(mode, n_devices, n_samples, wl.ctypes.data if len(wl) else None, len(wl), k, arr[0].ctypes.data,
                                 arr[1].ctypes.data, arr[2].ctypes.data, arr[3].ctypes.data, n.value, C.byref(n), C.byref(used)))
    return [tuple(int(a[i]) for a in arr) for i in range(n.value)], used.value


def screen_sharded(kmc_prefixes, k: int, text: bytes, win_begin, win_len, min_count: int = 1, devices=(0,),
                   mode: int = SHARD_AUTO) -> tuple[np.ndarray, ShardStats]:
    """kcfx_screen_sharded: every listed device screens its share (samples or window ranges); array [samples, windows]."""
    wb = np.ascontiguousarray(win_begin, dtype=np.int64)
    wl = np.ascontiguousarray(win_len, dtype=np.int32)
    buf = np.frombuffer(text, dtype=np.uint8)
    dev = (C.c_int32 * len(devices))(*devices)
    ctx = C.c_void_p()
    _check(lib().kcfx_ctx_create(dev, len(devices), C.byref(ctx)))
    try:
        names = [os.fsencode(p) for p in kmc_prefixes]
        arr = (C.c_char_p * max(len(names), 1))(*names)
        out = np.zeros((len(names), len(wb)), dtype=RESULT_DTYPE)
        st = ShardStats()
        _check(lib().kcfx_screen_sharded(ctx, arr, len(names), k, buf.ctypes.data if len(buf) else None, len(buf),
                                         wb.ctypes.data, wl.ctypes.data, len(wb), min_count, mode, out.ctypes.data, C.byref(st)))
    finally:
        lib().kcfx_ctx_destroy(ctx)
    return out, st


class KmcDb:
    """A KMC database re-indexed into an HBM-resident table (replaces `new KMC(prefix, inMemory)`, KMC.java:56)."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle
        self.info = DbInfo()
        _check(lib().kcfx_db_info(self._h, C.byref(self.info)))

    @classmethod
    def open(cls, kmc_prefix: str, device: int = 0, expected_queries: int = 0, no_fallback: bool = False,
             n_slices: int = 1, slice: int = 0) -> "KmcDb":
        """expected_queries = reference k-mers x passes this handle will serve (0 = unknown: the sparse, fastest table).
        n_slices > 1: the handle holds slice `slice` of the database's table (kcfx_db_open_slice): screened with
        Reference.screen_partial, merged with Reference.merge_from / merge_partial_async, reduced by Reference.finish."""
        h = C.c_void_p()
        if n_slices != 1:
            opts = DbOpts(int(expected_queries), int(no_fallback), 0)
            _check(lib().kcfx_db_open_slice(os.fsencode(kmc_prefix), device, C.byref(opts), int(n_slices), int(slice), C.byref(h)))
        elif expected_queries or no_fallback:
            opts = DbOpts(int(expected_queries), int(no_fallback), 0)
            _check(lib().kcfx_db_open_opts(os.fsencode(kmc_prefix), device, C.byref(opts), C.byref(h)))
        else:
            _check(lib().kcfx_db_open(os.fsencode(kmc_prefix), device, C.byref(h)))
        return cls(h)

    @classmethod
    def build(cls, kmers: np.ndarray, counts: np.ndarray, k: int, counter_size: int = 2,
              both_strands: bool = True, device: int = 0) -> "KmcDb":
        km = np.ascontiguousarray(kmers, dtype=np.uint64)
        ct = np.ascontiguousarray(counts, dtype=np.uint32)
        assert len(km) == len(ct)
        h = C.c_void_p()
        _check(lib().kcfx_db_build(km.ctypes.data, ct.ctypes.data, len(km), k, counter_size, int(both_strands), device, C.byref(h)))
        return cls(h)

    @property
    def k(self) -> int:
        return self.info.kmer_length

    def lookup(self, kmers: np.ndarray) -> np.ndarray:
        """Batched KMC.getCount (KMC.java:292-326)."""
        km = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(len(km), dtype=np.int32)
        _check(lib().kcfx_db_lookup(self._h, km.ctypes.data, len(km), out.ctypes.data))
        return out

    def screen(self, text: bytes, win_begin, win_len, min_count: int = 1) -> np.ndarray:
        """One-shot kcfx_screen on host buffers."""
        wb = np.ascontiguousarray(win_begin, dtype=np.int64)
        wl = np.ascontiguousarray(win_len, dtype=np.int32)
        buf = np.frombuffer(text, dtype=np.uint8)
        out = np.zeros(len(wb), dtype=RESULT_DTYPE)
        _check(lib().kcfx_screen(self._h, buf.ctypes.data if len(buf) else None, len(buf), wb.ctypes.data, wl.ctypes.data,
                                 len(wb), min_count, out.ctypes.data))
        return out

    def close(self) -> None:
        if self._h:
            lib().kcfx_db_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Cohort:
    """Up to 8 samples re-indexed into ONE cell table with a counter per sample and cell (kcfx_cohort_open): what the reference
    does as S getVariations runs + `cohort` (run_kcftools.sh:124-160, Cohort.java:71-119), located once per reference k-mer."""

    def __init__(self, kmc_prefixes, device: int = 0, expected_queries: int = 0):
        names = [os.fsencode(p) for p in kmc_prefixes]
        arr = (C.c_char_p * max(len(names), 1))(*names)
        opts = DbOpts(int(expected_queries), 0, 0)
        self._h = C.c_void_p()
        _check(lib().kcfx_cohort_open(arr, len(names), device, C.byref(opts), C.byref(self._h)))
        self.info = CohortInfo()
        _check(lib().kcfx_cohort_info(self._h, C.byref(self.info)))
        self.n_samples = len(names)

    def screen(self, ref: "Reference", min_count: int = 1) -> np.ndarray:
        """array [n_samples, n_windows]; each row equals Reference.screen on that sample's own database"""
        out = np.zeros((self.n_samples, ref.n_windows), dtype=RESULT_DTYPE)
        _check(lib().kcfx_cohort_screen_ref(self._h, ref._h, min_count, out.ctypes.data))
        return out

    def screen_async(self, ref: "Reference", out_device_ptr: int, stream: int = 0, min_count: int = 1) -> None:
        _check(lib().kcfx_cohort_screen_ref_async(self._h, ref._h, min_count, C.c_void_p(out_device_ptr), C.c_void_p(stream)))

    def close(self) -> None:
        if self._h:
            lib().kcfx_cohort_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Reference:
    """Reference text + window list resident in HBM (GetVariants.java:117-121 windows, :188-197 window text)."""

    def __init__(self, text: bytes, win_begin, win_len, k: int, device: int = 0, expected_screens: int = 0):
        """expected_screens = 1: ONE database will be screened against this reference -- it stays text only and the screening
        launch plans its tiles itself (kcfx_ref_opts); 0: planned once for any number of passes."""
        wb = np.ascontiguousarray(win_begin, dtype=np.int64)
        wl = np.ascontiguousarray(win_len, dtype=np.int32)
        assert len(wb) == len(wl)
        buf = np.frombuffer(text, dtype=np.uint8)
        self._h = C.c_void_p()
        opts = RefOpts(int(expected_screens), 0)
        _check(lib().kcfx_ref_create_opts(device, k, buf.ctypes.data if len(buf) else None, len(buf), wb.ctypes.data,
                                          wl.ctypes.data, len(wb), C.byref(opts), C.byref(self._h)))
        self.n_windows = len(wb)
        self.n_kmer_slots = lib().kcfx_ref_n_kmer_slots(self._h)
        self.device = device

    def screen(self, db: KmcDb, min_count: int = 1) -> np.ndarray:
        out = np.zeros(self.n_windows, dtype=RESULT_DTYPE)
        _check(lib().kcfx_screen_ref(db._h, self._h, min_count, out.ctypes.data))
        return out

    def screen_async(self, db: KmcDb, out_device_ptr: int, stream: int = 0, min_count: int = 1) -> None:
        """Asynchronous launch on `stream` (a hipStream_t as int); results stay in device memory."""
        _check(lib().kcfx_screen_ref_async(db._h, self._h, min_count, C.c_void_p(out_device_ptr), C.c_void_p(stream)))

    def screen_multi(self, dbs, min_count: int = 1) -> np.ndarray:
        """One pass over the reference for several resident databases: array [len(dbs), n_windows]."""
        handles = (C.c_void_p * max(len(dbs), 1))(*[C.cast(d._h, C.c_void_p).value for d in dbs])
        out = np.zeros((len(dbs), self.n_windows), dtype=RESULT_DTYPE)
        _check(lib().kcfx_screen_ref_multi(handles, len(dbs), self._h, min_count, out.ctypes.data))
        return out

    def screen_multi_async(self, dbs, out_device_ptr: int, stream: int = 0, min_count: int = 1) -> None:
        handles = (C.c_void_p * max(len(dbs), 1))(*[C.cast(d._h, C.c_void_p).value for d in dbs])
        _check(lib().kcfx_screen_ref_multi_async(handles, len(dbs), self._h, min_count, C.c_void_p(out_device_ptr),
                                                 C.c_void_p(stream)))

    # ---- a table partitioned over devices (KmcDb.open(..., n_slices, slice)) ----
    def screen_partial(self, db: "KmcDb", min_count: int = 1) -> None:
        """The screening kernel against one slice (or a whole table): the tiles' bitmaps and count sums stay in this reference."""
        _check(lib().kcfx_screen_ref_partial(db._h, self._h, min_count))

    def screen_partial_async(self, db: "KmcDb", stream: int = 0, min_count: int = 1) -> None:
        _check(lib().kcfx_screen_ref_partial_async(db._h, self._h, min_count, C.c_void_p(stream)))

    def merge_from(self, other: "Reference") -> None:
        """another slice's results (a reference of the same text and windows, any device of this process) into this one's"""
        _check(lib().kcfx_ref_merge_from(self._h, other._h))

    def partial_bytes(self) -> tuple[int, int]:
        b, c = C.c_int64(0), C.c_int64(0)
        lib().kcfx_ref_partial_bytes(self._h, C.byref(b), C.byref(c))
        return b.value, c.value

    def export_partial_async(self, bitmaps_device_ptr: int, sums_device_ptr: int, stream: int = 0) -> None:
        _check(lib().kcfx_ref_export_partial_async(self._h, C.c_void_p(bitmaps_device_ptr), C.c_void_p(sums_device_ptr), C.c_void_p(stream)))

    def merge_partial_async(self, bitmaps_device_ptr: int, sums_device_ptr: int, stream: int = 0) -> None:
        _check(lib().kcfx_ref_merge_partial_async(self._h, C.c_void_p(bitmaps_device_ptr), C.c_void_p(sums_device_ptr), C.c_void_p(stream)))

    def import_partial_async(self, bitmaps_device_ptr: int, sums_device_ptr: int, stream: int = 0) -> None:
        """device buffers of the export's layout INTO this reference's scratch (a rank that owns a window range starts its merge here)"""
        _check(lib().kcfx_ref_import_partial_async(self._h, C.c_void_p(bitmaps_device_ptr), C.c_void_p(sums_device_ptr), C.c_void_p(stream)))

    def tile_range(self, w_lo: int, w_hi: int) -> tuple[int, int]:
        """windows [w_lo, w_hi) of this reference are its tiles [t_lo, t_hi)"""
        a, b = C.c_int32(0), C.c_int32(0)
        _check(lib().kcfx_ref_tile_range(self._h, int(w_lo), int(w_hi), C.byref(a), C.byref(b)))
        return a.value, b.value

    def finish(self) -> np.ndarray:
        """reduce_windows over what this reference holds now (one slice's results, or the merged ones)"""
        out = np.zeros(self.n_windows, dtype=RESULT_DTYPE)
        _check(lib().kcfx_ref_finish(self._h, out.ctypes.data))
        return out

    def finish_async(self, out_device_ptr: int, stream: int = 0) -> None:
        _check(lib().kcfx_ref_finish_async(self._h, C.c_void_p(out_device_ptr), C.c_void_p(stream)))

    def reserve_multi(self, n_samples: int) -> None:
        """Allocate the multi-sample scratch now (needed before capturing screen_multi_async into a HIP graph)."""
        _check(lib().kcfx_ref_reserve_multi(self._h, n_samples))

    def prof_enable(self, on: bool = True) -> None:
        _check(lib().kcfx_prof_enable(self._h, int(on)))

    def prof_read(self, reset: bool = True) -> tuple[float, int]:
        ms, n = C.c_double(0), C.c_int64(0)
        _check(lib().kcfx_prof_read(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def plan_async(self, stream: int = 0) -> None:
        """The sample-independent half of a pass (2-bit pack, validity, minimizer choice) over the resident text again."""
        _check(lib().kcfx_ref_plan_async(self._h, C.c_void_p(stream)))

    @property
    def planned(self) -> bool:
        return bool(lib().kcfx_ref_is_planned(self._h))

    def prof_read_plan(self, reset: bool = True) -> tuple[float, int]:
        ms, n = C.c_double(0), C.c_int64(0)
        _check(lib().kcfx_prof_read_plan(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def close(self) -> None:
        if self._h:
            lib().kcfx_ref_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def score(res: np.ndarray, wi: float = 0.3, wt: float = 0.3, wr: float = 0.4) -> tuple[np.ndarray, np.ndarray]:
    """Data.update + Data.computeScore (Data.java:87,95-107) through the C ABI. Raises on bad weights."""
    r = np.ascontiguousarray(res, dtype=RESULT_DTYPE)
    sc = np.zeros(len(r), dtype=np.float64)
    md = np.zeros(len(r), dtype=np.float64)
    _check(lib().kcfx_score(r.ctypes.data, len(r), wi, wt, wr, sc.ctypes.data, md.ctypes.data))
    return sc, md


def windows(seq_len: int, window: int, step: int, k: int) -> tuple[np.ndarray, np.ndarray]:
    """GetVariants.getWindows, "window" case (GetVariants.java:293-320)."""
    n = C.c_int64(0)
    _check(lib().kcfx_windows(seq_len, window, step, k, None, None, 0, C.byref(n)))
    s = np.zeros(n.value, dtype=np.int32)
    e = np.zeros(n.value, dtype=np.int32)
    _check(lib().kcfx_windows(seq_len, window, step, k, s.ctypes.data, e.ctypes.data, n.value, C.byref(n)))
    return s, e
