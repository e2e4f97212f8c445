"""vstree_amd -- ctypes binding of the MI355X-native Vmengine query path.

The product is the C-ABI library vstree_amd/libvstree_amd.so (HIP kernels for
gfx950 + C host code; ABI in include/vstree_amd.h).  This module only maps it
into Python for the tests and bench.py, keeping the reference's operator names
(findcompletematches / findquerymatches / findmaximaluniquematches of
/root/reference/src/Vmengine/vmengineexport.h:4-81).

There is no CPU fallback: if the library is missing, importing fails; if no
GPU is present, every compute call returns the HIP error.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, "libvstree_amd.so")
# same code without a link-time dependency on libamdhip64 (see csrc/Makefile)
LIBPATH_NORT = os.path.join(_HERE, "libvstree_amd_nort.so")


def _loaded_hip_runtime():
    """path of a libamdhip64 already mapped into this process, if any (e.g.
    the one a PyTorch wheel bundles under torch/lib)"""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None

OWN_HIP_RUNTIME = "/opt/rocm/lib/libamdhip64.so"

SEPARATOR = 255
WILDCARD = 254
NO_SUBST = 0xFFFFFFFF

MATCH_DTYPE = np.dtype([("length", "<u8"), ("dbstart", "<u8"),
                        ("queryseq", "<u8"), ("querystart", "<u8")])
# vsa_match16: dbstart << 24 | length, queryseq << 16 | querystart
MATCH16_DTYPE = np.dtype([("dbstart_length", "<u8"),
                          ("queryseq_querystart", "<u8")])


def expand_match16(a):
    out = np.empty(len(a), MATCH_DTYPE)
    out["length"] = a["dbstart_length"] & np.uint64(0xFFFFFF)
    out["dbstart"] = a["dbstart_length"] >> np.uint64(24)
    out["queryseq"] = a["queryseq_querystart"] >> np.uint64(16)
    out["querystart"] = a["queryseq_querystart"] & np.uint64(0xFFFF)
    return out


class VsaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("vstree_amd error %d: %s" % (code, message))
        self.code = code
        self.message = message


class Tables(C.Structure):
    _fields_ = [("totallength", C.c_uint64), ("prefixlength", C.c_uint32),
                ("numofchars", C.c_uint32), ("integersize", C.c_uint32),
                ("largelcpvalues", C.c_uint64), ("tis", C.c_void_p),
                ("suf", C.c_void_p), ("lcp", C.c_void_p),
                ("llv", C.c_void_p), ("bck", C.c_void_p),
                ("bwt", C.c_void_p), ("querysepposition", C.c_uint64),
                ("hasindexedqueries", C.c_int)]


class IndexInfo(C.Structure):
    _fields_ = [("totallength", C.c_uint64), ("numofcodes", C.c_uint64),
                ("largelcpvalues", C.c_uint64), ("device_bytes", C.c_uint64),
                ("prefixlength", C.c_uint32), ("numofchars", C.c_uint32),
                ("device_integersize", C.c_uint32), ("device", C.c_int),
                ("hasindexedqueries", C.c_int), ("hasbwt", C.c_int),
                ("deepprefix", C.c_uint32)]


class QueriesInfo(C.Structure):
    _fields_ = [("numofqueries", C.c_uint64), ("numofsymbols", C.c_uint64),
                ("minlength", C.c_uint64), ("maxlength", C.c_uint64),
                ("offset", C.c_uint64), ("device", C.c_int)]


class SinkParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("palindromic", C.c_int),
                ("selfpalindromic", C.c_int), ("showmode", C.c_uint32), ("numofchars", C.c_uint32),
                ("threads", C.c_int),
                ("leastlength", C.c_uint64), ("totallength", C.c_uint64),
                ("numofsequences", C.c_uint64), ("markpos", C.c_void_p),
                ("numofquerysequences", C.c_uint64),
                ("totalquerylength", C.c_uint64),
                ("numofqueries", C.c_uint64),
                ("querytotallength", C.c_uint64),
                ("querystart", C.c_void_p), ("querylength", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sumlength", C.c_uint64),
                ("searches", C.c_uint64), ("candidates", C.c_uint64),
                ("search_kernel_ms", C.c_double),
                ("total_device_ms", C.c_double), ("anchor_ms", C.c_double),
                ("kernel_searches", C.c_uint64),
                ("first_kernel_ms", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


PROCESSMATCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


def _load():
    global LIBPATH
    if not os.path.exists(LIBPATH):
        raise ImportError(
            "%s is missing: build it with `make -C vstree_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback"
            % LIBPATH)
    # One HIP runtime per process.  VSTREE_AMD_RUNTIME says which:
    #   own   libvstree_amd.so, which links /opt/rocm/lib/libamdhip64.so
    #   host  libvstree_amd_nort.so: the same objects, bound to the runtime the
    #         process has mapped already (a PyTorch wheel's, under torch/lib)
    #   auto  (default) host if a runtime is mapped at import time, own if not
    # -- the explicit values are for callers that do not want the outcome to
    # depend on the order of their imports: each fails where it cannot hold.
    choice = os.environ.get("VSTREE_AMD_RUNTIME", "auto")
    if choice not in ("auto", "own", "host"):
        raise ImportError("VSTREE_AMD_RUNTIME=%r: expected own, host or auto"
                          % choice)
    hip = _loaded_hip_runtime()
    if choice == "host" and hip is None:
        raise ImportError("VSTREE_AMD_RUNTIME=host, but no HIP runtime is "
                          "mapped in this process yet (import torch first)")
    if choice == "own" and hip is not None and \
            os.path.realpath(hip) != os.path.realpath(OWN_HIP_RUNTIME):
        raise ImportError("VSTREE_AMD_RUNTIME=own, but this process has "
                          "mapped %s: two HIP runtimes do not share a device"
                          % hip)
    if choice == "host" or (choice == "auto" and hip is not None and
                            os.path.exists(LIBPATH_NORT)):
        C.CDLL(hip, mode=C.RTLD_GLOBAL)
        LIBPATH = LIBPATH_NORT
    lib = C.CDLL(LIBPATH)
    V, U64, U32, I = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    PP = C.POINTER(C.c_void_p)
    sig = {
        "vsa_messagespace": (C.c_char_p, []),
        "vsa_index_from_tables": (I, [C.POINTER(Tables), I, PP]),
        "vsa_index_open": (I, [C.c_char_p, I, PP]),
        "vsa_index_close": (None, [V]),
        "vsa_index_getinfo": (I, [V, C.POINTER(IndexInfo)]),
        "vsa_index_build": (I, [V, U64, U32, U32, I, PP]),
        "vsa_index_build_device": (I, [V, U64, U32, U32, I, PP]),
        "vsa_index_download": (I, [V, V, V, V, V, V, V]),
        "vsa_index_set_queryseparator": (I, [V, U64]),
        "vsa_index_set_queryspeedup": (I, [V, U32]),
        "vsa_index_clone": (I, [V, I, PP]),
        "vsa_pipeline_open": (I, [V, I, U64, U32, U64, PP]),
        "vsa_pipeline_hostbuffer": (V, [V]),
        "vsa_pipeline_open_packed": (I, [V, I, U64, U32, U64, U64, PP]),
        "vsa_pipeline_hostrows": (I, [V, PP, PP]),
        "vsa_pipeline_submit_packed": (I, [V, U64, U64]),
        "vsa_pipeline_submit": (I, [V, U64]),
        "vsa_pipeline_next": (I, [V, PP, C.POINTER(U64)]),
        "vsa_pipeline_finish": (I, [V, PP, C.POINTER(U64),
                                    C.POINTER(Stats)]),
        "vsa_pipeline_finish16": (I, [V, PP, C.POINTER(U64),
                                      C.POINTER(Stats)]),
        "vsa_pipeline_close": (None, [V]),
        "vsa_pipeline_set_offset": (I, [V, U64]),
        "vsa_pipeline_take_candidates": (I, [V, PP, C.POINTER(U64),
                                             C.POINTER(U32)]),
        "vsa_rows_partition_device": (I, [V, U64, U32, U32, I, U64, I, V,
                                          V]),
        "vsa_mkvtree": (I, [C.POINTER(C.c_char_p), U32, C.POINTER(C.c_char_p),
                            U32, C.c_char_p, U32, U32, I, I]),
        "vsa_queries_from_host": (I, [V, U64, V, V, U64, I, PP]),
        "vsa_queries_from_device": (I, [V, U64, U32, I, PP]),
        "vsa_packed_words": (U32, [U32]),
        "vsa_pack_reads": (I, [V, U64, U32, U64, V, V, U64, C.POINTER(U64)]),
        "vsa_pack_reads_mt": (I, [V, U64, U32, U64, V, V, U64, C.POINTER(U64),
                                  U32]),
        "vsa_queries_from_host_packed": (I, [V, U64, U32, V, U64, I, PP]),
        "vsa_queries_reverse_complement": (I, [V, PP]),
        "vsa_queries_free": (None, [V]),
        "vsa_queries_set_offset": (I, [V, U64]),
        "vsa_queries_getinfo": (I, [V, C.POINTER(QueriesInfo)]),
        "vsa_result_count": (U64, [V]),
        "vsa_result_getstats": (I, [V, C.POINTER(Stats)]),
        "vsa_result_fetch": (I, [V, V, U64]),
        "vsa_result_device_matches": (V, [V]),
        "vsa_result_free": (None, [V]),
        "vsa_result_copy_device": (I, [V, V, U64]),
        "vsa_mumuniqueinquery": (I, [V, U64, I, PP]),
        "vsa_mumuniqueinquery_range": (I, [V, U64, I, U64, PP]),
        "vsa_index_make_sti1": (I, [V, V]),
        "vsa_findcompletematches": (I, [V, V, PP]),
        "vsa_findapproxcompletematches": (I, [V, V, I, U64, I, PP]),
        "vsa_findapproxcompletematches_cb": (I, [V, V, I, U64, I,
                                                 PROCESSMATCH, V]),
        "vsa_findquerymatches": (I, [V, V, I, I, U64, PP]),
        "vsa_findmumcandidates": (I, [V, V, U64, I, PP]),
        "vsa_result_partition": (I, [V, U32, U64, V, V, V]),
        "vsa_result_partition_own": (I, [V, U32, I, U64, V, V, V]),
        "vsa_result_partition_device": (I, [V, U32, I, U64, V, V]),
        "vsa_findmaximaluniquematches": (I, [V, U64, PP]),
        "vsa_findmaximaluniquematches_range": (I, [V, U64, U64, U64, PP]),
        "vsa_findmaximalrepeats": (I, [V, U64, PP]),
        "vsa_findmaximalrepeats_cb": (I, [V, U64, PROCESSMATCH, V]),
        "vsa_findsupermaximalrepeats": (I, [V, U64, PP]),
        "vsa_findsupermaximalrepeats_cb": (I, [V, U64, PROCESSMATCH, V]),
        "vsa_findtandems": (I, [V, U64, PP]),
        "vsa_findmumcandidates_packed": (I, [V, V, U64, C.c_uint32, PP]),
        "vsa_findmumcandidates_grouped": (
            I, [V, V, U64, C.c_uint32, U32, I, V, U64, V, PP]),
        "vsa_result_packbits": (C.c_uint32, [V]),
        "vsa_mumuniqueinquery_range_packed": (
            I, [V, U64, C.c_uint32, U64, I, U64, PP]),
        "vsa_mumuniqueinquery_range_packed2": (
            I, [V, U64, V, U64, C.c_uint32, U64, I, U64, PP]),
        "vsa_findtandems_cb": (I, [V, U64, PROCESSMATCH, V]),
        "vsa_findcompletematches_cb": (I, [V, V, PROCESSMATCH, V]),
        "vsa_findquerymatches_cb": (I, [V, V, I, I, U64, PROCESSMATCH, V]),
        "vsa_findmaximaluniquematches_cb": (I, [V, U64, PROCESSMATCH, V]),
        "vsa_measure_random_read": (I, [U64, I, I, C.POINTER(C.c_double)]),
        "vsa_measure_table_read": (I, [V, I, I, C.POINTER(C.c_double)]),
        "vsa_sink_open": (I, [C.POINTER(SinkParams), PP]),
        "vsa_sink_close": (None, [V]),
        "vsa_sink_format": (C.c_int64, [V, V, U64, V, U64]),
        "vsa_sink_write": (I, [V, V, U64, V]),
        "vsa_splitmix64_at": (U64, [U64, U64]),
        "vsa_synth_genome": (None, [U64, U64, V]),
        "vsa_synth_query_plan": (None, [U64, U64, U64, U32, V, V, V]),
        "vsa_synth_queries": (None, [U64, V, U64, U64, U32, V, V]),
        "vsa_synth_genome_device": (I, [U64, U64, V, I]),
        "vsa_synth_queries_device": (I, [V, U64, V, V, V, U64, U32, V, I]),
        "vsa_device_malloc": (I, [U64, I, PP]),
        "vsa_device_free": (I, [V, I]),
        "vsa_device_upload": (I, [V, V, U64, I]),
        "vsa_device_download": (I, [V, V, U64, I]),
        "vsa_device_count": (I, []),
        "vsa_device_synchronize": (I, [I]),
        "vsa_device_trim": (I, [I]),
        "vsa_device_meminfo": (I, [I, C.POINTER(U64), C.POINTER(U64)]),
        "vsa_measure_stream_read": (I, [U64, I, C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib, sorted(sig)


lib, ABI_SYMBOLS = _load()


def messagespace():
    return lib.vsa_messagespace().decode(errors="replace")


def _check(rc):
    if rc != 0:
        raise VsaError(rc, messagespace())


def _ptr(a):
    return None if a is None else a.ctypes.data


class Index:
    """An enhanced suffix array resident in one GPU's HBM (vsa_index)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_tables(cls, totallength, prefixlength, numofchars, tis, suf, lcp,
                    llv, bck, bwt=None, querysepposition=0,
                    hasindexedqueries=False, device=0):
        suf = np.ascontiguousarray(suf)
        assert suf.dtype in (np.uint32, np.uint64)
        tis = np.ascontiguousarray(tis, np.uint8)
        lcp = np.ascontiguousarray(lcp, np.uint8)
        llv = np.ascontiguousarray(llv, suf.dtype)
        bck = np.ascontiguousarray(bck, suf.dtype)
        bwt = None if bwt is None else np.ascontiguousarray(bwt, np.uint8)
        t = Tables(int(totallength), int(prefixlength), int(numofchars),
                   suf.dtype.itemsize * 8, llv.shape[0] // 2, _ptr(tis),
                   _ptr(suf), _ptr(lcp), _ptr(llv), _ptr(bck), _ptr(bwt),
                   int(querysepposition), int(bool(hasindexedqueries)))
        h = C.c_void_p()
        _check(lib.vsa_index_from_tables(C.byref(t), device, C.byref(h)))
        return cls(h)

    @classmethod
    def open(cls, indexname, device=0):
        """mapvirtualtreeifyoucan + upload (vsa_index_open)."""
        h = C.c_void_p()
        _check(lib.vsa_index_open(os.fsencode(indexname), device,
                                  C.byref(h)))
        return cls(h)

    @classmethod
    def build(cls, tis, numofchars=4, prefixlength=0, device=0):
        tis = np.ascontiguousarray(tis, np.uint8)
        h = C.c_void_p()
        _check(lib.vsa_index_build(_ptr(tis), tis.shape[0], numofchars,
                                   prefixlength, device, C.byref(h)))
        return cls(h)

    @classmethod
    def build_device(cls, device_tis, totallength, numofchars=4,
                     prefixlength=0, device=0):
        h = C.c_void_p()
        _check(lib.vsa_index_build_device(device_tis, totallength, numofchars,
                                          prefixlength, device, C.byref(h)))
        return cls(h)

    def info(self):
        i = IndexInfo()
        _check(lib.vsa_index_getinfo(self._h, C.byref(i)))
        return i

    def download(self, with_bwt=True):
        """-> dict of numpy arrays with the device tables."""
        i = self.info()
        dt = np.uint32 if i.device_integersize == 32 else np.uint64
        n = i.totallength
        out = {"tis": np.zeros(n, np.uint8), "suf": np.zeros(n + 1, dt),
               "lcp": np.zeros(n + 1, np.uint8),
               "llv": np.zeros(2 * i.largelcpvalues, dt),
               "bck": np.zeros(2 * i.numofcodes, dt),
               "bwt": (np.zeros(n + 1, np.uint8)
                       if (with_bwt and i.hasbwt) else None)}
        _check(lib.vsa_index_download(self._h, _ptr(out["tis"]),
                                      _ptr(out["suf"]), _ptr(out["lcp"]),
                                      _ptr(out["llv"]), _ptr(out["bck"]),
                                      _ptr(out["bwt"])))
        return out

    def clone(self, device=0):
        """replica on another device: device-to-device copies, no rebuild"""
        h = C.c_void_p()
        _check(lib.vsa_index_clone(self._h, int(device), C.byref(h)))
        return Index(h)

    def set_queryseparator(self, pos):
        _check(lib.vsa_index_set_queryseparator(self._h, int(pos)))

    def set_queryspeedup(self, level):
        """vmatch -qspeedup: 0 or 2 (default), the order of MEM lists."""
        _check(lib.vsa_index_set_queryspeedup(self._h, int(level)))

    def make_sti1(self):
        out = np.zeros(self.info().totallength + 1, np.uint8)
        _check(lib.vsa_index_make_sti1(self._h, _ptr(out)))
        return out

    def close(self):
        if self._h and lib is not None:   # None at interpreter shutdown
            lib.vsa_index_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Queries:
    """A batch of query sequences resident in HBM (vsa_queries)."""

    def __init__(self, handle, nq):
        self._h = handle
        self.nq = nq

    @classmethod
    def from_host(cls, symbols, start, length, device=0):
        symbols = np.ascontiguousarray(symbols, np.uint8)
        start = np.ascontiguousarray(start, np.uint64)
        length = np.ascontiguousarray(length, np.uint64)
        h = C.c_void_p()
        _check(lib.vsa_queries_from_host(_ptr(symbols), symbols.shape[0],
                                         _ptr(start), _ptr(length),
                                         start.shape[0], device, C.byref(h)))
        return cls(h, start.shape[0])

    @classmethod
    def from_host_packed(cls, symbols, m, device=0, stride=None):
        """nq reads of m mapped symbols (read i at symbols[i * stride:]) ->
        packed on the host (vsa_pack_reads), uploaded as rows + side list"""
        symbols = np.ascontiguousarray(symbols, np.uint8)
        stride = m if stride is None else stride
        nq = 0 if len(symbols) < m else (len(symbols) - m) // stride + 1
        rows, special, ns = pack_reads(symbols, nq, m, stride)
        h = C.c_void_p()
        _check(lib.vsa_queries_from_host_packed(
            _ptr(rows), nq, m, _ptr(special), ns, device, C.byref(h)))
        return cls(h, nq)

    @classmethod
    def from_device(cls, device_symbols, nq, m, device=0):
        h = C.c_void_p()
        _check(lib.vsa_queries_from_device(device_symbols, nq, m, device,
                                           C.byref(h)))
        return cls(h, nq)

    def set_offset(self, offset):
        _check(lib.vsa_queries_set_offset(self._h, int(offset)))

    def info(self):
        qi = QueriesInfo()
        _check(lib.vsa_queries_getinfo(self._h, C.byref(qi)))
        return qi

    def reverse_complement(self):
        """vmatch -p: every sequence reversed and complemented on its own"""
        h = C.c_void_p()
        _check(lib.vsa_queries_reverse_complement(self._h, C.byref(h)))
        return Queries(h, self.nq)

    def close(self):
        if self._h and lib is not None:
            lib.vsa_queries_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Result:
    """A match list resident in HBM (vsa_result)."""

    def __init__(self, handle):
        self._h = handle

    @property
    def count(self):
        return int(lib.vsa_result_count(self._h))

    def stats(self):
        s = Stats()
        _check(lib.vsa_result_getstats(self._h, C.byref(s)))
        return s

    def copy_device(self, device_ptr, capacity):
        _check(lib.vsa_result_copy_device(self._h, device_ptr, capacity))

    def partition(self, nparts, totallength, device_ptr, own=-1):
        """records grouped by the index range of their dbstart, written to
        device_ptr; -> (number of records, largest right end) per part.
        own >= 0: that part lies behind all others"""
        counts = np.zeros(nparts, np.uint64)
        maxright = np.zeros(nparts, np.uint64)
        _check(lib.vsa_result_partition_own(self._h, int(nparts), int(own),
                                            int(totallength), device_ptr,
                                            _ptr(counts), _ptr(maxright)))
        return counts, maxright

    def partition_device(self, nparts, totallength, device_ptr, device_meta,
                         own=-1):
        """partition() that leaves (counts, largest right ends) in device
        memory at device_meta (2*nparts uint64) and does not wait for the
        GPU"""
        _check(lib.vsa_result_partition_device(
            self._h, int(nparts), int(own), int(totallength), device_ptr,
            device_meta))

    @property
    def packbits(self):
        """length bits of a packed candidate result, 0 for records"""
        return int(lib.vsa_result_packbits(self._h))

    @property
    def rowwords(self):
        """8-byte words per row that partition() writes"""
        return 2 if self.packbits else 4

    def fetch(self):
        n = self.count
        out = np.zeros(n, MATCH_DTYPE)
        if n:
            _check(lib.vsa_result_fetch(self._h, _ptr(out), n))
        return out

    def close(self):
        if self._h and lib is not None:
            lib.vsa_result_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def findcompletematches(index, queries):
    """vmatch -complete -q (Vmengine/fcomplete.c:263).  On the reference's
    short-query error the VsaError carries the matches found before it in
    .partial."""
    h = C.c_void_p()
    rc = lib.vsa_findcompletematches(index._h, queries._h, C.byref(h))
    res = Result(h) if h else None
    if rc != 0:
        e = VsaError(rc, messagespace())
        e.partial = res
        raise e
    return res


NOT_COVERED = -4


def findapproxcompletematches(index, queries, doedist, distvalue,
                              percent=False):
    """vmatch -complete -e K | -h K -q (Vmengine/approxcompl.c:138); the
    distance of a match travels in its querystart field.  VsaError.code ==
    NOT_COVERED: a configuration the engine leaves to the CPU reference."""
    h = C.c_void_p()
    rc = lib.vsa_findapproxcompletematches(index._h, queries._h,
                                           int(doedist), int(distvalue),
                                           int(percent), C.byref(h))
    res = Result(h) if h else None
    if rc != 0:
        e = VsaError(rc, messagespace())
        e.partial = res
        raise e
    return res


def findquerymatches(index, queries, searchlength, mum=False, cand=False,
                     speedup=None):
    """vmatch [-mum [cand]] -l L -q (Vmengine/fquery.c:1009); speedup = the
    -qspeedup level (0 or 2) to set on the index first."""
    if speedup is not None:
        index.set_queryspeedup(speedup)
    h = C.c_void_p()
    _check(lib.vsa_findquerymatches(index._h, queries._h, int(mum),
                                    int(cand), int(searchlength),
                                    C.byref(h)))
    return Result(h)


def findmumcandidates(index, queries, searchlength, ordered=True):
    """vmatch -mum cand; ordered=False: as the kernel left them (for a
    filter that sorts them anyway)"""
    h = C.c_void_p()
    _check(lib.vsa_findmumcandidates(index._h, queries._h, int(searchlength),
                                     int(ordered), C.byref(h)))
    return Result(h)


def findmumcandidates_packed(index, queries, searchlength, lengthbits=0):
    """the candidates as (sort key, value) pairs for the multi-GPU filter,
    see the header"""
    h = C.c_void_p()
    _check(lib.vsa_findmumcandidates_packed(index._h, queries._h,
                                            int(searchlength),
                                            int(lengthbits), C.byref(h)))
    return Result(h)


def findmumcandidates_grouped(index, queries, searchlength, lengthbits,
                              nparts, own, device_rows, capacity,
                              device_meta):
    """findmumcandidates_packed + Result.partition_device in one call ->
    (result, grouped): grouped False = more than `capacity` rows, nothing
    was written to device_rows"""
    h = C.c_void_p()
    rc = lib.vsa_findmumcandidates_grouped(
        index._h, queries._h, int(searchlength), int(lengthbits),
        int(nparts), int(own), device_rows, int(capacity), device_meta,
        C.byref(h))
    if rc not in (0, 1):
        _check(rc)
    return Result(h), rc == 0


def findmaximalrepeats(index, searchlength):
    """vmatch -l L IDX (Vmengine/vmatfind.c:487), the reference's order."""
    h = C.c_void_p()
    _check(lib.vsa_findmaximalrepeats(index._h, int(searchlength),
                                      C.byref(h)))
    return Result(h)


def findsupermaximalrepeats(index, searchlength):
    """vmatch -supermax -l L IDX (Vmengine/fsuper.c:142)."""
    h = C.c_void_p()
    _check(lib.vsa_findsupermaximalrepeats(index._h, int(searchlength),
                                           C.byref(h)))
    return Result(h)


def findtandems(index, searchlength):
    """vmatch -tandem -l L IDX (Vmengine/ftandem.c:261)."""
    h = C.c_void_p()
    _check(lib.vsa_findtandems(index._h, int(searchlength), C.byref(h)))
    return Result(h)


def findmaximaluniquematches(index, searchlength, first=None, last=None):
    """vmatch -mum -l L IDX (Vmengine/fmumself.c:10); first/last: the part
    first <= i < last of the reference's scan (multi-GPU form)."""
    h = C.c_void_p()
    if first is None and last is None:
        _check(lib.vsa_findmaximaluniquematches(index._h, int(searchlength),
                                                C.byref(h)))
    else:
        _check(lib.vsa_findmaximaluniquematches_range(
            index._h, int(searchlength), int(first or 0),
            int(last if last is not None else 2 ** 64 - 1), C.byref(h)))
    return Result(h)


def mumuniqueinquery(device_candidates, ncandidates, device=0):
    """kurtz/cleanMUMcand.c:55 on candidates resident in device memory."""
    h = C.c_void_p()
    _check(lib.vsa_mumuniqueinquery(device_candidates, int(ncandidates),
                                    device, C.byref(h)))
    return Result(h)


def mumuniqueinquery_range(device_candidates, ncandidates, carry_dbright,
                           device=0):
    """the filter on one dbstart range (multi-GPU), see the header"""
    h = C.c_void_p()
    _check(lib.vsa_mumuniqueinquery_range(device_candidates,
                                          int(ncandidates), device,
                                          int(carry_dbright), C.byref(h)))
    return Result(h)


def mumuniqueinquery_range_packed(device_rows, nrows, lengthbits, totallength,
                                  carry_dbright, device=0):
    """the filter on one dbstart range of packed candidate rows"""
    h = C.c_void_p()
    _check(lib.vsa_mumuniqueinquery_range_packed(
        device_rows, int(nrows), int(lengthbits), int(totallength), device,
        int(carry_dbright), C.byref(h)))
    return Result(h)


def mumuniqueinquery_range_packed2(device_rows, nrows, more_rows, nmore,
                                   lengthbits, totallength, carry_dbright,
                                   device=0):
    """the filter on one dbstart range of packed candidate rows that lie in
    two places (own rows, received rows)"""
    h = C.c_void_p()
    _check(lib.vsa_mumuniqueinquery_range_packed2(
        device_rows, int(nrows), more_rows, int(nmore), int(lengthbits),
        int(totallength), device, int(carry_dbright), C.byref(h)))
    return Result(h)


def _collector(stop_after=None):
    got = []

    def cb(info, mptr):
        m = np.frombuffer((C.c_uint64 * 4).from_address(mptr), np.uint64)
        got.append(tuple(int(x) for x in m))
        if stop_after is not None and len(got) >= stop_after:
            return 1
        return 0
    return got, PROCESSMATCH(cb)


def findcompletematches_cb(index, queries, stop_after=None):
    got, cb = _collector(stop_after)
    rc = lib.vsa_findcompletematches_cb(index._h, queries._h, cb, None)
    return rc, got


def findapproxcompletematches_cb(index, queries, doedist, distvalue,
                                 percent=False, stop_after=None):
    got, cb = _collector(stop_after)
    rc = lib.vsa_findapproxcompletematches_cb(
        index._h, queries._h, int(doedist), int(distvalue), int(percent), cb,
        None)
    return rc, got


def findquerymatches_cb(index, queries, searchlength, mum=False, cand=False,
                        stop_after=None, speedup=None):
    if speedup is not None:
        index.set_queryspeedup(speedup)
    got, cb = _collector(stop_after)
    rc = lib.vsa_findquerymatches_cb(index._h, queries._h, int(mum),
                                     int(cand), int(searchlength), cb, None)
    return rc, got


def findmaximaluniquematches_cb(index, searchlength, stop_after=None):
    got, cb = _collector(stop_after)
    rc = lib.vsa_findmaximaluniquematches_cb(index._h, int(searchlength), cb,
                                             None)
    return rc, got


class Pipeline:
    """vsa_pipeline_*: host memory in, host memory out, three batches in
    flight.  mode: 0 -complete, 1 MEM, 2 -mum cand, 3 -mum."""

    def __init__(self, index, mode, searchlength, querylength, maxqueries,
                 packed=False, maxspecial=None):
        self._h = None
        h = C.c_void_p()
        self.packed = bool(packed)
        self.maxspecial = int(maxqueries if maxspecial is None
                              else maxspecial)
        if packed:
            _check(lib.vsa_pipeline_open_packed(
                index._h, int(mode), int(searchlength), int(querylength),
                int(maxqueries), self.maxspecial, C.byref(h)))
        else:
            _check(lib.vsa_pipeline_open(index._h, int(mode),
                                         int(searchlength), int(querylength),
                                         int(maxqueries), C.byref(h)))
        self._h, self._index = h, index
        self.m, self.maxqueries = int(querylength), int(maxqueries)
        self.W = int(lib.vsa_packed_words(self.m))

    def hostrows(self):
        """packed pipelines: (rows, special) numpy views of the page-locked
        room of the next batch, or None when all batches are in flight"""
        r, sp = C.c_void_p(), C.c_void_p()
        rc = lib.vsa_pipeline_hostrows(self._h, C.byref(r), C.byref(sp))
        if rc == 1:
            return None
        _check(rc)
        rows = np.ctypeslib.as_array(
            (C.c_uint64 * (self.W * self.maxqueries)).from_address(r.value))
        special = np.ctypeslib.as_array(
            (C.c_uint8 * max(1, self.m * self.maxspecial)).from_address(
                sp.value))
        return rows, special

    def pack_into_slot(self, symbols, nq, stride=None):
        """packs nq reads into the next slot and submits it; False when all
        batches are in flight"""
        got = self.hostrows()
        if got is None:
            return False
        rows, special = got
        ns = C.c_uint64(0)
        symbols = np.ascontiguousarray(symbols, np.uint8)
        _check(lib.vsa_pack_reads(_ptr(symbols), nq, self.m,
                                  self.m if stride is None else stride,
                                  _ptr(rows), _ptr(special), self.maxspecial,
                                  C.byref(ns)))
        _check(lib.vsa_pipeline_submit_packed(self._h, int(nq),
                                              int(ns.value)))
        return True

    def hostbuffer(self):
        """numpy view of the page-locked buffer of the next batch, or None
        when all batches are in flight"""
        p = lib.vsa_pipeline_hostbuffer(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(
            (C.c_uint8 * (self.m * self.maxqueries)).from_address(p))

    def submit(self, nq):
        _check(lib.vsa_pipeline_submit(self._h, int(nq)))

    def next(self, copy=True):
        """-> (rc, matches): rc 1 = nothing outstanding"""
        ptr, n = C.c_void_p(), C.c_uint64()
        rc = lib.vsa_pipeline_next(self._h, C.byref(ptr), C.byref(n))
        if rc == 1 or n.value == 0:
            return rc, np.zeros(0, MATCH_DTYPE)
        a = np.ctypeslib.as_array(
            (C.c_uint64 * (4 * n.value)).from_address(ptr.value)).view(
                MATCH_DTYPE)
        return rc, a.copy() if copy else a

    def finish(self, copy=True, compact=False):
        """compact: vsa_pipeline_finish16 -> an array of MATCH16_DTYPE
        (expand_match16 gives the records)"""
        ptr, n, st = C.c_void_p(), C.c_uint64(), Stats()
        if compact:
            _check(lib.vsa_pipeline_finish16(self._h, C.byref(ptr),
                                             C.byref(n), C.byref(st)))
            if n.value == 0:
                return np.zeros(0, MATCH16_DTYPE), st
            a = np.ctypeslib.as_array(
                (C.c_uint64 * (2 * n.value)).from_address(ptr.value)).view(
                    MATCH16_DTYPE)
            return (a.copy() if copy else a), st
        _check(lib.vsa_pipeline_finish(self._h, C.byref(ptr), C.byref(n),
                                       C.byref(st)))
        if n.value == 0:
            return np.zeros(0, MATCH_DTYPE), st
        a = np.ctypeslib.as_array(
            (C.c_uint64 * (4 * n.value)).from_address(ptr.value)).view(
                MATCH_DTYPE)
        return (a.copy() if copy else a), st

    def close(self):
        if self._h and lib is not None:
            lib.vsa_pipeline_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


# ---- host match sink ------------------------------------------------------

SINK_COMPLETE, SINK_QUERY, SINK_SELF, SINK_APPROX_EDIST, \
    SINK_APPROX_HAMMING = range(5)
SHOW_ABSOLUTE, SHOW_NODIST, SHOW_NOEVALUE, SHOW_NOSCORE, SHOW_NOIDENTITY = (
    1, 2, 4, 8, 16)


class Sink:
    """vmatch's output lines for match records (vsa_sink, host side)."""

    def __init__(self, kind, totallength, markpos, numofchars=4,
                 querystart=None, querylength=None, querytotallength=0,
                 numofquerysequences=0, totalquerylength=0, leastlength=0,
                 palindromic=False, showmode=0, threads=0,
                 selfpalindromic=False):
        self._keep = [np.ascontiguousarray(markpos, np.uint64)]
        p = SinkParams()
        p.kind, p.palindromic = int(kind), int(bool(palindromic))
        p.selfpalindromic = int(bool(selfpalindromic))
        p.showmode, p.numofchars = int(showmode), int(numofchars)
        p.threads = int(threads)
        p.leastlength, p.totallength = int(leastlength), int(totallength)
        p.numofsequences = self._keep[0].shape[0] + 1
        p.markpos = _ptr(self._keep[0]) if self._keep[0].shape[0] else None
        p.numofquerysequences = int(numofquerysequences)
        p.totalquerylength = int(totalquerylength)
        if querystart is not None:
            qs = np.ascontiguousarray(querystart, np.uint64)
            ql = np.ascontiguousarray(querylength, np.uint64)
            self._keep += [qs, ql]
            p.numofqueries = qs.shape[0]
            p.querytotallength = int(querytotallength)
            p.querystart, p.querylength = _ptr(qs), _ptr(ql)
        self._h = C.c_void_p()
        _check(lib.vsa_sink_open(C.byref(p), C.byref(self._h)))

    def format(self, matches):
        """matches: MATCH_DTYPE array -> bytes (one line per match)"""
        matches = np.ascontiguousarray(matches, MATCH_DTYPE)
        cap = 192 * (matches.shape[0] + 1)
        buf = np.empty(cap, np.uint8)
        n = lib.vsa_sink_format(self._h, _ptr(matches), matches.shape[0],
                                _ptr(buf), cap)
        if n < 0:
            raise VsaError(int(n), messagespace())
        return buf[:n].tobytes()

    def close(self):
        if self._h and lib is not None:
            lib.vsa_sink_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


# ---- synthetic inputs ------------------------------

GENOME_SEED = 42
QUERY_SEED = 4242


def synth_genome(n, seed=GENOME_SEED):
    g = np.zeros(n, np.uint8)
    lib.vsa_synth_genome(seed, n, _ptr(g))
    return g


def synth_queries(genome, nq, m, seed=QUERY_SEED):
    genome = np.ascontiguousarray(genome, np.uint8)
    q = np.zeros(nq * m, np.uint8)
    lib.vsa_synth_queries(seed, _ptr(genome), genome.shape[0], nq, m,
                          _ptr(q), None)
    return q


def pack_reads(symbols, nq, m, stride=None, specialcap=None, threads=1):
    """-> (rows u64[nq * W], special u8[ns * m], ns)"""
    stride = m if stride is None else stride
    W = int(lib.vsa_packed_words(m))
    rows = np.zeros(nq * W, np.uint64)
    cap = nq if specialcap is None else specialcap
    special = np.zeros(max(cap, 1) * m, np.uint8)
    ns = C.c_uint64(0)
    _check(lib.vsa_pack_reads_mt(_ptr(symbols), nq, m, stride, _ptr(rows),
                                 _ptr(special), cap, C.byref(ns), threads))
    return rows, special[:ns.value * m], int(ns.value)


def synth_query_plan(n, nq, m, seed=QUERY_SEED):
    pos = np.zeros(nq, np.uint64)
    sub = np.zeros(nq, np.uint32)
    step = np.zeros(nq, np.uint8)
    lib.vsa_synth_query_plan(seed, n, nq, m, _ptr(pos), _ptr(sub),
                             _ptr(step))
    return pos, sub, step


def mkvtree(dbfiles, indexname, queryfiles=(), prefixlength=0, integersize=64,
            withskp=True, device=0):
    """mkvtree -db .. [-q ..] -indexname .. -dna -pl -allout on the GPU."""
    db = (C.c_char_p * len(dbfiles))(*[os.fsencode(f) for f in dbfiles])
    qf = (C.c_char_p * max(1, len(queryfiles)))(
        *[os.fsencode(f) for f in queryfiles])
    _check(lib.vsa_mkvtree(db, len(dbfiles), qf, len(queryfiles),
                           os.fsencode(indexname), prefixlength, integersize,
                           int(withskp), device))


def device_count():
    return int(lib.vsa_device_count())


def device_synchronize(device=0):
    _check(lib.vsa_device_synchronize(device))


def device_malloc(nbytes, device=0):
    p = C.c_void_p()
    _check(lib.vsa_device_malloc(nbytes, device, C.byref(p)))
    return p


def device_upload(dptr, array, device=0):
    array = np.ascontiguousarray(array)
    _check(lib.vsa_device_upload(dptr, _ptr(array), array.nbytes, device))


def device_download(array, dptr, device=0):
    """device memory -> a contiguous numpy array (its size decides)"""
    assert array.flags["C_CONTIGUOUS"]
    _check(lib.vsa_device_download(_ptr(array), dptr, array.nbytes, device))


def device_meminfo(device=0, trim=True):
    """(free, total) bytes of the device, the library's cache handed back"""
    if trim:
        _check(lib.vsa_device_trim(device))
    f, t = C.c_uint64(0), C.c_uint64(0)
    _check(lib.vsa_device_meminfo(device, C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def device_free(p, device=0):
    _check(lib.vsa_device_free(p, device))


def measure_stream_read(nbytes=1 << 30, device=0):
    g = C.c_double()
    _check(lib.vsa_measure_stream_read(nbytes, device, C.byref(g)))
    return g.value
