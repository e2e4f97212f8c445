"""ctypes mirror of include/vstree_amd_multi.h (tests only: the product
surface is the C ABI, which integration/vmengine_shim.c binds)."""
import ctypes as C
import os

import numpy as np

import vstree_amd as V

LIBPATH = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                       "libvstree_amd_multi.so")
COMPLETE, MEM, MUMCAND, MUM = 0, 1, 2, 3
MATCH_DTYPE = np.dtype([("length", "<u8"), ("dbstart", "<u8"),
                        ("queryseq", "<u8"), ("querystart", "<u8")])

_V, _I, _U32, _U64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
_PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "vsa_multi_from_tables": (_I, [_V, C.POINTER(C.c_int), _U32, _PP]),
    "vsa_multi_replicate": (_I, [_V, C.POINTER(C.c_int), _U32, _PP]),
    "vsa_multi_ndevices": (_U32, [_V]),
    "vsa_multi_index": (_V, [_V, _U32]),
    "vsa_multi_close": (None, [_V]),
    "vsa_multi_uses_rccl": (_I, [_V]),
    "vsa_multi_findmatches": (_I, [_V, _I, _U64, _V, _U64, _V, _V, _U64, _PP,
                                   C.POINTER(_U64), C.POINTER(V.Stats)]),
    "vsa_multi_free_matches": (None, [_V]),
    "vsa_multi_findmatches_device": (_I, [_V, _I, _U64, _PP, _PP,
                                          C.POINTER(V.Stats)]),
    "vsa_multi_pipeline_open": (_I, [_V, _I, _U64, _U32, _U64, _U64, _PP]),
    "vsa_multi_pipeline_hostrows": (_I, [_V, _PP, _PP]),
    "vsa_multi_pipeline_submit": (_I, [_V, _U64, _U64]),
    "vsa_multi_pipeline_next": (_I, [_V, _PP, C.POINTER(_U64)]),
    "vsa_multi_pipeline_finish": (_I, [_V, _PP, C.POINTER(_U64),
                                       C.POINTER(V.Stats)]),
    "vsa_multi_pipeline_close": (None, [_V]),
    "vsa_multi_findmatches_cb": (_I, [_V, _I, _U64, _V, _U64, _V, _V, _U64,
                                      V.PROCESSMATCH, _V]),
    "vsa_multi_findapproxcompletematches": (
        _I, [_V, _I, _U64, _I, _V, _U64, _V, _V, _U64, _PP, C.POINTER(_U64),
             C.POINTER(V.Stats)]),
    "vsa_multi_findapproxcompletematches_cb": (
        _I, [_V, _I, _U64, _I, _V, _U64, _V, _V, _U64, V.PROCESSMATCH, _V]),
}


def _load():
    if not os.path.exists(LIBPATH):
        raise ImportError("%s is missing: make -C vstree_amd/csrc" % LIBPATH)
    # libvstree_amd.so is mapped already (import vstree_amd): the multi
    # library binds to that copy -- which has to be the one it was linked
    # with.  In a process that carries another HIP runtime (torch) the mirror
    # has loaded libvstree_amd_nort.so, and this library would bring
    # libvstree_amd.so and a second runtime in beside it.
    if V.LIBPATH != os.path.join(os.path.dirname(LIBPATH),
                                 "libvstree_amd.so"):
        raise ImportError(
            "vstree_amd.multi drives the GPUs through the HIP runtime "
            "libvstree_amd.so links; this process has mapped another one (%s "
            "is loaded): use it from a process without torch "
            "(VSTREE_AMD_RUNTIME=own makes that explicit)" % V.LIBPATH)
    lib = C.CDLL(LIBPATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class Multi:
    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_tables(cls, totallength, prefixlength, numofchars, tis, suf, lcp,
                    llv, bck, bwt=None, devices=(0,)):
        suf = np.ascontiguousarray(suf)
        keep = [np.ascontiguousarray(tis, np.uint8), suf,
                np.ascontiguousarray(lcp, np.uint8),
                np.ascontiguousarray(llv, suf.dtype),
                np.ascontiguousarray(bck, suf.dtype),
                None if bwt is None else np.ascontiguousarray(bwt, np.uint8)]
        t = V.Tables(int(totallength), int(prefixlength), int(numofchars),
                     suf.dtype.itemsize * 8, keep[3].shape[0] // 2,
                     V._ptr(keep[0]), V._ptr(keep[1]), V._ptr(keep[2]),
                     V._ptr(keep[3]), V._ptr(keep[4]), V._ptr(keep[5]), 0, 0)
        h = C.c_void_p()
        d = (C.c_int * len(devices))(*devices)
        V._check(lib.vsa_multi_from_tables(C.byref(t), d, len(devices),
                                           C.byref(h)))
        return cls(h)

    @classmethod
    def replicate(cls, index, devices):
        """takes the index over (it becomes replica 0)"""
        h = C.c_void_p()
        d = (C.c_int * len(devices))(*devices)
        V._check(lib.vsa_multi_replicate(index._h, d, len(devices),
                                         C.byref(h)))
        index._h = None
        return cls(h)

    def ndevices(self):
        return lib.vsa_multi_ndevices(self._h)

    def uses_rccl(self):
        return bool(lib.vsa_multi_uses_rccl(self._h))

    def set_queryspeedup(self, level):
        for r in range(self.ndevices()):
            V._check(V.lib.vsa_index_set_queryspeedup(
                lib.vsa_multi_index(self._h, r), int(level)))

    def findmatches(self, mode, symbols, start, length, searchlength=0):
        """-> (matches, Stats, rc, message); rc != 0 keeps the matches the
        reference would have delivered before the error"""
        symbols = np.ascontiguousarray(symbols, np.uint8)
        start = np.ascontiguousarray(start, np.uint64)
        length = np.ascontiguousarray(length, np.uint64)
        out, n, st = C.c_void_p(), C.c_uint64(), V.Stats()
        rc = lib.vsa_multi_findmatches(
            self._h, int(mode), int(searchlength), V._ptr(symbols),
            symbols.shape[0], V._ptr(start), V._ptr(length), start.shape[0],
            C.byref(out), C.byref(n), C.byref(st))
        m = np.zeros(n.value, MATCH_DTYPE)
        if n.value:
            C.memmove(m.ctypes.data, out.value, n.value * 32)
        lib.vsa_multi_free_matches(out)
        return m, st, rc, V.messagespace() if rc != 0 else ""

    def findmatches_device(self, mode, blocks, searchlength=0):
        """blocks[r]: a V.Queries on the device of replica r (offset set) ->
        ([V.Result per replica], Stats of the job, rc, message); the lists
        stay in HBM"""
        n = self.ndevices()
        assert len(blocks) == n
        qs = (C.c_void_p * n)(*[b._h for b in blocks])
        rs = (C.c_void_p * n)()
        st = V.Stats()
        rc = lib.vsa_multi_findmatches_device(
            self._h, int(mode), int(searchlength), qs, rs, C.byref(st))
        msg = V.messagespace() if rc != 0 else ""
        res = [V.Result(C.c_void_p(rs[r])) if rs[r] else None
               for r in range(n)]
        return res, st, rc, msg

    def findapproxcompletematches(self, symbols, start, length, doedist,
                                  distvalue, percent=False):
        """vmatch -complete -e K | -h K over all replicas -> (matches, Stats,
        rc, message); the distance of a match in its querystart field"""
        symbols = np.ascontiguousarray(symbols, np.uint8)
        start = np.ascontiguousarray(start, np.uint64)
        length = np.ascontiguousarray(length, np.uint64)
        out, n, st = C.c_void_p(), C.c_uint64(), V.Stats()
        rc = lib.vsa_multi_findapproxcompletematches(
            self._h, int(doedist), int(distvalue), int(percent),
            V._ptr(symbols), symbols.shape[0], V._ptr(start), V._ptr(length),
            start.shape[0], C.byref(out), C.byref(n), C.byref(st))
        m = np.zeros(n.value, MATCH_DTYPE)
        if n.value:
            C.memmove(m.ctypes.data, out.value, n.value * 32)
        lib.vsa_multi_free_matches(out)
        return m, st, rc, V.messagespace() if rc != 0 else ""

    def findmatches_cb(self, mode, symbols, start, length, searchlength=0,
                       stop_after=None):
        symbols = np.ascontiguousarray(symbols, np.uint8)
        start = np.ascontiguousarray(start, np.uint64)
        length = np.ascontiguousarray(length, np.uint64)
        got, cb = V._collector(stop_after)
        rc = lib.vsa_multi_findmatches_cb(
            self._h, int(mode), int(searchlength), V._ptr(symbols),
            symbols.shape[0], V._ptr(start), V._ptr(length), start.shape[0],
            cb, None)
        return rc, got

    def close(self):
        if self._h and lib is not None:
            lib.vsa_multi_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


class MultiPipeline:
    """vsa_multi_pipeline_*: a packed pipeline per replica, the batches of a
    job dealt out in turn and delivered in submission order"""

    def __init__(self, multi, mode, searchlength, querylength, maxqueries,
                 maxspecial=None):
        self._h = None
        h = C.c_void_p()
        self.maxspecial = int(maxqueries if maxspecial is None
                              else maxspecial)
        V._check(lib.vsa_multi_pipeline_open(
            multi._h, int(mode), int(searchlength), int(querylength),
            int(maxqueries), self.maxspecial, C.byref(h)))
        self._h, self._multi = h, multi
        self.m, self.maxqueries = int(querylength), int(maxqueries)
        self.W = int(V.lib.vsa_packed_words(self.m))
        self.n = multi.ndevices()

    def hostrows(self):
        r, sp = C.c_void_p(), C.c_void_p()
        rc = lib.vsa_multi_pipeline_hostrows(self._h, C.byref(r), C.byref(sp))
        if rc == 1:
            return None
        V._check(rc)
        rows = np.ctypeslib.as_array(
            (C.c_uint64 * (self.W * self.maxqueries)).from_address(r.value))
        special = np.ctypeslib.as_array(
            (C.c_uint8 * max(1, self.m * self.maxspecial)).from_address(
                sp.value))
        return rows, special

    def submit(self, nq, nspecial):
        V._check(lib.vsa_multi_pipeline_submit(self._h, int(nq),
                                               int(nspecial)))

    def pack_into_slot(self, symbols, nq, stride=None):
        got = self.hostrows()
        if got is None:
            return False
        rows, special = got
        ns = C.c_uint64(0)
        symbols = np.ascontiguousarray(symbols, np.uint8)
        V._check(V.lib.vsa_pack_reads(
            V._ptr(symbols), nq, self.m,
            self.m if stride is None else stride, V._ptr(rows),
            V._ptr(special), self.maxspecial, C.byref(ns)))
        self.submit(nq, ns.value)
        return True

    def next(self, copy=True):
        ptr, n = C.c_void_p(), C.c_uint64()
        rc = lib.vsa_multi_pipeline_next(self._h, C.byref(ptr), C.byref(n))
        if rc == 1 or n.value == 0:
            return rc, np.zeros(0, MATCH_DTYPE)
        a = np.ctypeslib.as_array(
            (C.c_uint64 * (4 * n.value)).from_address(ptr.value)).view(
                MATCH_DTYPE)
        return rc, a.copy() if copy else a

    def finish(self, copy=True):
        """-mum: ([list of replica r], Stats of the job)"""
        ptrs = (C.c_void_p * self.n)()
        cnts = (C.c_uint64 * self.n)()
        st = V.Stats()
        V._check(lib.vsa_multi_pipeline_finish(self._h, ptrs, cnts,
                                               C.byref(st)))
        out = []
        for r in range(self.n):
            if cnts[r] == 0:
                out.append(np.zeros(0, MATCH_DTYPE))
                continue
            a = np.ctypeslib.as_array(
                (C.c_uint64 * (4 * cnts[r])).from_address(ptrs[r])).view(
                    MATCH_DTYPE)
            out.append(a.copy() if copy else a)
        return out, st

    def close(self):
        if self._h and lib is not None:
            lib.vsa_multi_pipeline_close(self._h)
            self._h = None

    def __del__(self):
        self.close()
