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
    # library binds to that copy
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
