"""Shared test plumbing: the CPU oracle (oracle/liboracle.so) through ctypes, a
numpy reader for mkvtree index files, FASTA -> alphabet-mapped symbols, and
runners/parsers for the reference programs of oracle/_ref (when present).

Everything in here is test infrastructure; the product lives in vstree_amd/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden")
MKVTREE_REF = os.path.join(REF_DIR, "mkvtree_ref")
VMATCH_REF = os.path.join(REF_DIR, "vmatch_ref")

SEPARATOR = 255
WILDCARD = 254


def have_ref():
    return os.access(MKVTREE_REF, os.X_OK) and os.access(VMATCH_REF, os.X_OK)


# --------------------------------------------------------------------------
# oracle binding
# --------------------------------------------------------------------------

class OrcIndex(C.Structure):
    _fields_ = [("n", C.c_uint64), ("prefixlength", C.c_uint32),
                ("numofchars", C.c_uint32), ("isize", C.c_uint32),
                ("nllv", C.c_uint64), ("tis", C.c_void_p),
                ("suf", C.c_void_p), ("lcp", C.c_void_p),
                ("llv", C.c_void_p), ("bck", C.c_void_p),
                ("bwt", C.c_void_p), ("sti1", C.c_void_p),
                ("querysepposition", C.c_uint64), ("hasqueries", C.c_int)]


class OrcMatches(C.Structure):
    _fields_ = [("m", C.c_void_p), ("n", C.c_uint64), ("cap", C.c_uint64)]


class OrcCounters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("charcomp", "sufprobes", "lcpreads", "bckreads", "searches",
                 "emitted")]


MATCH_DTYPE = np.dtype([("length", "<u8"), ("dbstart", "<u8"),
                        ("queryseq", "<u8"), ("querystart", "<u8")])


def build_oracle():
    """Compile oracle/vsoracle.c if the .so is missing or stale."""
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f)
            for f in ("vsoracle.c", "vsindex.c", "vsapprox.c", "vsself.c",
                      "vsoracle_body.inc", "vsoracle.h")]
    if (not os.path.exists(so) or
            os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs)):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", srcs[0],
                               srcs[1], srcs[2], srcs[3], "-lm", "-o", so])
    return so


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        lib.orc_matches_init.argtypes = [C.POINTER(OrcMatches)]
        lib.orc_matches_free.argtypes = [C.POINTER(OrcMatches)]
        lib.orc_counters_get.argtypes = [C.POINTER(OrcCounters)]
        common = [C.POINTER(OrcIndex), C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_uint64]
        lib.orc_findcompletematches.argtypes = common + [
            C.POINTER(OrcMatches), C.c_char_p]
        lib.orc_findcompletematches_online.argtypes = common + [
            C.POINTER(OrcMatches), C.c_char_p]
        lib.orc_findquerymatches.argtypes = common + [
            C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(OrcMatches),
            C.c_char_p]
        lib.orc_findmaximaluniquematches.argtypes = [
            C.POINTER(OrcIndex), C.c_uint64, C.POINTER(OrcMatches),
            C.c_char_p]
        lib.orc_findapproxcompletematches.argtypes = common + [
            C.c_int, C.c_uint64, C.c_int, C.POINTER(OrcMatches), C.c_char_p]
        lib.orc_findsupermax.argtypes = [
            C.POINTER(OrcIndex), C.c_uint64, C.POINTER(OrcMatches),
            C.c_char_p]
        lib.orc_findmaximalrepeats.argtypes = [
            C.POINTER(OrcIndex), C.c_uint64, C.POINTER(OrcMatches),
            C.c_char_p]
        lib.orc_findtandems.argtypes = [
            C.POINTER(OrcIndex), C.c_uint64, C.POINTER(OrcMatches),
            C.c_char_p]
        lib.orc_getoptsplit.argtypes = [C.c_int] + [C.c_uint64] * 5
        lib.orc_getoptsplit.restype = C.c_uint64
        lib.orc_mumuniqueinquery.argtypes = [C.c_void_p, C.c_uint64,
                                             C.POINTER(OrcMatches)]
        lib.orc_mumuniqueinquery_carry.argtypes = [C.c_void_p, C.c_uint64,
                                                   C.c_uint64,
                                                   C.POINTER(OrcMatches)]
        lib.orc_recommendedprefixlength.argtypes = [C.c_uint32, C.c_uint64]
        lib.orc_recommendedprefixlength.restype = C.c_uint32
        lib.orc_build_tables.argtypes = [
            C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
            C.c_void_p]
        lib.orc_build_tables.restype = C.c_int64
        _oracle = lib
    return _oracle


class OracleError(RuntimeError):
    pass


class Index:
    """Tables of one mkvtree index as numpy arrays (host memory)."""

    def __init__(self, n, prefixlength, numofchars, tis, suf, lcp, llv, bck,
                 bwt=None, sti1=None, ssp=None, numofsequences=1,
                 querysepposition=0, hasqueries=False, prj=None):
        self.n = int(n)
        self.prefixlength = int(prefixlength)
        self.numofchars = int(numofchars)
        self.tis = np.ascontiguousarray(tis, dtype=np.uint8)
        self.suf = np.ascontiguousarray(suf)
        self.lcp = np.ascontiguousarray(lcp, dtype=np.uint8)
        self.llv = np.ascontiguousarray(llv, dtype=self.suf.dtype)
        self.bck = np.ascontiguousarray(bck, dtype=self.suf.dtype)
        self.bwt = None if bwt is None else np.ascontiguousarray(bwt, np.uint8)
        self.sti1 = None if sti1 is None else np.ascontiguousarray(sti1,
                                                                   np.uint8)
        self.ssp = (np.zeros(0, np.uint64) if ssp is None
                    else np.asarray(ssp, dtype=np.uint64))
        self.numofsequences = int(numofsequences)
        self.querysepposition = int(querysepposition)
        self.hasqueries = bool(hasqueries)
        self.prj = prj or {}
        assert self.suf.dtype in (np.uint32, np.uint64)
        assert self.suf.shape[0] == self.n + 1
        assert self.lcp.shape[0] == self.n + 1
        assert self.tis.shape[0] == self.n

    @property
    def isize(self):
        return self.suf.dtype.itemsize

    @property
    def nllv(self):
        return self.llv.shape[0] // 2

    def as_width(self, bits):
        """Same index with 32- or 64-bit suf/bck/llv entries."""
        dt = np.uint32 if bits == 32 else np.uint64
        return Index(self.n, self.prefixlength, self.numofchars, self.tis,
                     self.suf.astype(dt), self.lcp, self.llv.astype(dt),
                     self.bck.astype(dt), self.bwt, self.sti1, self.ssp,
                     self.numofsequences, self.querysepposition,
                     self.hasqueries, self.prj)

    def orc(self):
        p = lambda a: None if a is None else a.ctypes.data
        return OrcIndex(self.n, self.prefixlength, self.numofchars,
                        self.isize, self.nllv, p(self.tis), p(self.suf),
                        p(self.lcp), p(self.llv), p(self.bck), p(self.bwt),
                        p(self.sti1), self.querysepposition,
                        int(self.hasqueries))

    def seq_rel(self, pos):
        """absolute position(s) -> (sequence number, relative position),
        like getseqinfo (kurtz-basic/multiseq-adv.c:277)."""
        pos = np.asarray(pos, dtype=np.uint64)
        if self.ssp.size == 0:
            return np.zeros_like(pos), pos
        seq = np.searchsorted(self.ssp, pos, side="right").astype(np.uint64)
        starts = np.concatenate(([0], self.ssp + 1)).astype(np.uint64)
        return seq, pos - starts[seq]


class Queries:
    """Query sequences: one symbol buffer + (start, length) per query."""

    def __init__(self, symbols, start, length, names=None):
        self.symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
        self.start = np.ascontiguousarray(start, dtype=np.uint64)
        self.length = np.ascontiguousarray(length, dtype=np.uint64)
        self.names = names
        assert self.start.shape == self.length.shape

    @property
    def nq(self):
        return self.start.shape[0]

    @staticmethod
    def from_list(seqs):
        """seqs: list of uint8 arrays/lists; stored with separators between
        them like a reference Multiseq."""
        buf, start, length, pos = [], [], [], 0
        for i, s in enumerate(seqs):
            s = np.asarray(s, dtype=np.uint8)
            if i > 0:
                buf.append(np.array([SEPARATOR], np.uint8))
                pos += 1
            start.append(pos)
            length.append(len(s))
            buf.append(s)
            pos += len(s)
        symbols = np.concatenate(buf) if buf else np.zeros(0, np.uint8)
        return Queries(symbols, start, length)

    @staticmethod
    def uniform(block, m):
        """block: (nq*m,) symbols, query i = block[i*m:(i+1)*m]."""
        block = np.ascontiguousarray(block, dtype=np.uint8)
        nq = block.shape[0] // m
        return Queries(block, np.arange(nq, dtype=np.uint64) * m,
                       np.full(nq, m, np.uint64))

    def seq(self, i):
        s = int(self.start[i])
        return self.symbols[s:s + int(self.length[i])]


def _take(out):
    n = int(out.n)
    if n == 0:
        arr = np.zeros(0, MATCH_DTYPE)
    else:
        buf = (C.c_uint8 * (n * MATCH_DTYPE.itemsize)).from_address(out.m)
        arr = np.frombuffer(buf, dtype=MATCH_DTYPE).copy()
    oracle_lib().orc_matches_free(C.byref(out))
    return arr


def _qargs(queries):
    return (queries.symbols.ctypes.data, queries.start.ctypes.data,
            queries.length.ctypes.data, queries.nq)


def oracle_complete(index, queries, online=False):
    lib = oracle_lib()
    out, err = OrcMatches(), C.create_string_buffer(512)
    lib.orc_matches_init(C.byref(out))
    oi = index.orc()
    fn = (lib.orc_findcompletematches_online if online
          else lib.orc_findcompletematches)
    rc = fn(C.byref(oi), *_qargs(queries), C.byref(out), err)
    res = _take(out)
    if rc != 0:
        e = OracleError(err.value.decode())
        e.partial = res
        raise e
    return res


def oracle_supermax(index, searchlength):
    """vmatch -supermax -l L IDX: (length, start1, start2, 0)"""
    lib = oracle_lib()
    out, err = OrcMatches(), C.create_string_buffer(512)
    lib.orc_matches_init(C.byref(out))
    oi = index.orc()
    rc = lib.orc_findsupermax(C.byref(oi), int(searchlength), C.byref(out),
                              err)
    res = _take(out)
    if rc != 0:
        raise OracleError(err.value.decode())
    return res


def oracle_repeats(index, searchlength):
    """vmatch -l L IDX (maximal repeats): (length, start1, start2, 0)"""
    lib = oracle_lib()
    out, err = OrcMatches(), C.create_string_buffer(512)
    lib.orc_matches_init(C.byref(out))
    oi = index.orc()
    rc = lib.orc_findmaximalrepeats(C.byref(oi), int(searchlength),
                                    C.byref(out), err)
    res = _take(out)
    if rc != 0:
        raise OracleError(err.value.decode())
    return res


def oracle_tandems(index, searchlength):
    """vmatch -tandem -l L IDX: (length, start, start + length, 0)"""
    lib = oracle_lib()
    out, err = OrcMatches(), C.create_string_buffer(512)
    lib.orc_matches_init(C.byref(out))
    oi = index.orc()
    rc = lib.orc_findtandems(C.byref(oi), int(searchlength), C.byref(out),
                             err)
    res = _take(out)
    if rc != 0:
        raise OracleError(err.value.decode())
    return res


class OracleNotCovered(OracleError):
    pass


def oracle_approx(index, queries, doedist, distvalue, percent=False):
    """vmatch -complete -e K | -h K: matches carry the distance in
    querystart.  percent: False / 0 absolute, True / 1 percent of the read's
    length (Kp), 2 "best of" (Kb)"""
    if int(percent) == 2:
        return _oracle_approx_bestof(index, queries, doedist, distvalue)
    lib = oracle_lib()
    out, err = OrcMatches(), C.create_string_buffer(512)
    lib.orc_matches_init(C.byref(out))
    oi = index.orc()
    rc = lib.orc_findapproxcompletematches(
        C.byref(oi), *_qargs(queries), int(doedist), int(distvalue),
        int(percent), C.byref(out), err)
    res = _take(out)
    if rc != 0:
        e = (OracleNotCovered if rc == -4 else OracleError)(
            err.value.decode())
        e.partial = res
        raise e
    return res


def _oracle_approx_bestof(index, queries, doedist, distvalue):
    """-e Kb / -h Kb (Vmengine/initcompl.c:59-77, approxcompl.c:80-122;
    fcomplete.c:251-252 restores K in front of every read): every read at the
    smallest threshold t <= m K / 100 at which it has a match; reads without
    one report nothing.  The existence check of the reference is monotone in
    t, so t = the read's smallest distance within m K / 100."""
    first = oracle_approx(index, queries, doedist, distvalue, percent=1)
    best = np.full(queries.nq, -1, np.int64)
    for qi, d in zip(first["queryseq"], first["querystart"]):
        if best[qi] < 0 or d < best[qi]:
            best[qi] = d
    parts = []
    for t in sorted(set(int(b) for b in best if b >= 0)):
        which = np.flatnonzero(best == t)
        sub = Queries(queries.symbols, queries.start[which],
                      queries.length[which])
        m = oracle_approx(index, sub, doedist, t).copy()
        m["queryseq"] = which[m["queryseq"]]
        parts.append(m)
    if not parts:
        return first[:0]
    allm = np.concatenate(parts)
    return allm[np.argsort(allm["queryseq"], kind="stable")]


def selfmum_scan_range(index, searchlength, first=2, last=None):
    """findmaximaluniquematches (Vmengine/fmumself.c:33-64) restated with numpy
    for the values first <= i < last of its loop variable: what one rank of
    the multi-GPU scan computes.  first=2, last=None is the whole loop."""
    n = index.n
    lcp = index.lcp.astype(np.int64)
    if index.nllv:
        llv = index.llv.reshape(-1, 2).astype(np.int64)
        lcp[llv[:, 0]] = llv[:, 1]
    first, last = max(2, int(first)), n if last is None else min(int(last), n)
    i = np.arange(first, max(first, last), dtype=np.int64)
    f, s, t = lcp[i - 2].copy(), lcp[i - 1], lcp[i]
    f[i == 2] = 0                        # firstlcp starts at 0
    ok = (s >= searchlength) & (f < s) & (t < s)
    i, s = i[ok], s[ok]
    a, b = index.suf[i - 2].astype(np.int64), index.suf[i - 1].astype(np.int64)
    s1, s2 = np.minimum(a, b), np.maximum(a, b)
    sep = index.querysepposition
    x, y = index.bwt[i - 1], index.bwt[i - 2]
    ok = (s1 < sep) & (s2 > sep) & ((s1 == 0) | (x >= WILDCARD) |
                                    (y >= WILDCARD) | (x != y))
    out = np.zeros(int(ok.sum()), MATCH_DTYPE)
    out["length"], out["dbstart"], out["queryseq"] = s[ok], s1[ok], s2[ok]
    return out


def sti1_from_tables(suf, lcp, prefixlength, chunk=1 << 26):
    """Table stitab1 from its definition (Mkvtree/mkvprocess.c:583-612):
    sti1[suf[i]] = min(255, i - start of i's run of lcp >= prefixlength);
    numpy, in chunks, for tables of any size."""
    n1 = len(suf)
    out = np.zeros(n1, np.uint8)
    carry = 0                               # start of the run reaching in
    for a in range(0, n1, chunk):
        b = min(n1, a + chunk)
        idx = np.arange(a, b, dtype=np.int64)
        starts = np.where(lcp[a:b] < prefixlength, idx, -1)
        if a == 0:
            starts[0] = 0
        runstart = np.maximum.accumulate(np.maximum(starts, carry))
        carry = int(runstart[-1])
        out[suf[a:b]] = np.minimum(idx - runstart, 255).astype(np.uint8)
    return out


def oracle_querymatches(index, queries, searchlength, mum=False, cand=False,
                        speedup=0):
    lib = oracle_lib()
    out, err = OrcMatches(), C.create_string_buffer(512)
    lib.orc_matches_init(C.byref(out))
    oi = index.orc()
    rc = lib.orc_findquerymatches(C.byref(oi), *_qargs(queries), int(mum),
                                  int(cand), int(searchlength), int(speedup),
                                  C.byref(out), err)
    res = _take(out)
    if rc != 0:
        raise OracleError(err.value.decode())
    return res


def oracle_selfmum(index, searchlength):
    lib = oracle_lib()
    out, err = OrcMatches(), C.create_string_buffer(512)
    lib.orc_matches_init(C.byref(out))
    oi = index.orc()
    rc = lib.orc_findmaximaluniquematches(C.byref(oi), int(searchlength),
                                          C.byref(out), err)
    res = _take(out)
    if rc != 0:
        raise OracleError(err.value.decode())
    return res


def oracle_mumfilter(cand, carry=0):
    """kurtz/cleanMUMcand.c:55-118 on a candidate array (copied)"""
    lib = oracle_lib()
    out = OrcMatches()
    lib.orc_matches_init(C.byref(out))
    c2 = np.ascontiguousarray(cand.copy())
    lib.orc_mumuniqueinquery_carry(c2.ctypes.data, len(c2), int(carry),
                                   C.byref(out))
    return _take(out)


def oracle_counters(reset=False):
    lib = oracle_lib()
    c = OrcCounters()
    lib.orc_counters_get(C.byref(c))
    if reset:
        lib.orc_counters_reset()
    return {k: int(getattr(c, k)) for k, _ in OrcCounters._fields_}


# --------------------------------------------------------------------------
# FASTA and alphabets
# --------------------------------------------------------------------------

def dna_map():
    """Symbol map of mkvtree -dna (the .al1 file it writes: aA cC gG tTuU,
    wildcard class nsywrkvbdhm in both cases)."""
    m = np.full(256, 253, np.uint8)          # 253 = not in the alphabet
    for code, chars in enumerate(("aA", "cC", "gG", "tTuU")):
        for ch in chars:
            m[ord(ch)] = code
    for ch in "nsywrkvbdhmNSYWRKVBDHM":
        m[ord(ch)] = WILDCARD
    return m


def read_fasta(path):
    """-> list of (description, bytes) records."""
    recs, desc, chunks = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if desc is not None:
                    recs.append((desc, b"".join(chunks)))
                desc, chunks = line[1:].decode(), []
            elif desc is not None:
                chunks.append(line.replace(b" ", b""))
    if desc is not None:
        recs.append((desc, b"".join(chunks)))
    return recs


def fasta_queries(path, symmap=None):
    symmap = dna_map() if symmap is None else symmap
    recs = read_fasta(path)
    seqs = [symmap[np.frombuffer(s, np.uint8)] for _, s in recs]
    for s in seqs:
        assert not (s == 253).any(), "symbol outside the alphabet"
    q = Queries.from_list(seqs)
    q.names = [d for d, _ in recs]
    return q


def write_fasta(path, records, width=60):
    """records: list of (description, uint8 code array or bytes of letters)."""
    letters = np.full(256, ord("n"), np.uint8)   # wildcard code 254 -> n
    letters[:4] = np.frombuffer(b"acgt", np.uint8)
    with open(path, "wb") as f:
        for desc, seq in records:
            if not isinstance(seq, (bytes, bytearray)):
                seq = letters[np.asarray(seq, np.uint8)].tobytes()
            f.write(b">" + desc.encode() + b"\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + b"\n")


def write_fasta_fast(path, records, width=60):
    """the same file as write_fasta, written line matrix by line matrix (a
    Python loop over the 1.7 M lines of a 100 Mbp record takes minutes)"""
    letters = np.full(256, ord("n"), np.uint8)
    letters[:4] = np.frombuffer(b"acgt", np.uint8)
    with open(path, "wb") as f:
        for desc, seq in records:
            seq = letters[np.asarray(seq, np.uint8)]
            f.write(b">" + desc.encode() + b"\n")
            k = (len(seq) // width) * width
            if k:
                rows = seq[:k].reshape(-1, width)
                nl = np.full((rows.shape[0], 1), 10, np.uint8)
                f.write(np.hstack([rows, nl]).tobytes())
            if k < len(seq):
                f.write(seq[k:].tobytes() + b"\n")


# Index-builder parity at the size SURVEY 8f-1 names: two texts whose index
# files are too large to keep -- only the md5 of every file the reference's
# mkvtree wrote is (tests/golden/bigindex.json, scripts/make_golden_big.py).
# The texts come out of generators, here, so that the GPU test rebuilds the
# very same FASTA files on the box.
BIG_CASES = ("c100m", "r20m")


def big_case_records(case):
    """-> list of (description, codes) of the FASTA file of a big case"""
    import vstree_amd as V
    if case == "c100m":
        # the synthetic genome of SURVEY 8d at 100 Mbp: uniform bases, one
        # record
        return [("synthetic_genome seed=42", V.synth_genome(100000000))]
    if case == "r20m":
        # 200 sequences, 20 Mbp: a 5 kb unit planted 300 times with a few
        # substitutions each (lcp values in the thousands: llv), an exact
        # tandem array, runs of wildcards, sequences of different lengths
        rng = np.random.default_rng(20240)
        lens = rng.integers(40000, 160000, 200)
        lens = (lens * (20000000 / lens.sum())).astype(np.int64)
        unit = rng.integers(0, 4, 5000).astype(np.uint8)
        recs = []
        for i, ln in enumerate(lens):
            t = rng.integers(0, 4, int(ln)).astype(np.uint8)
            for r in range(int(rng.integers(0, 4))):
                if ln > 6000:
                    p = int(rng.integers(0, ln - 5000))
                    u = unit.copy()
                    for e in range(int(rng.integers(0, 5))):
                        u[int(rng.integers(0, 5000))] = rng.integers(0, 4)
                    t[p:p + 5000] = u
            if i % 17 == 3:
                p = int(rng.integers(0, ln - 4000))
                t[p:p + 3700] = np.tile(unit[:37], 100)
            for r in range(int(rng.integers(0, 3))):
                p = int(rng.integers(0, ln - 600))
                t[p:p + int(rng.integers(1, 500))] = WILDCARD
            recs.append(("seq%d planted repeats, wildcards len=%d" % (i, ln),
                         t))
        return recs
    raise KeyError(case)


# --------------------------------------------------------------------------
# mkvtree index files
# --------------------------------------------------------------------------

def read_prj(path):
    prj = {}
    with open(path) as f:
        for line in f:
            k, _, v = line.strip().partition("=")
            if k in ("dbfile", "queryfile"):
                prj.setdefault(k, []).append(v)
            else:
                prj[k] = int(v)
    return prj


def load_mkvtree_index(prefix, numofchars=4):
    """Read indexname.{prj,tis,suf,lcp,llv,bck,bwt,sti1,ssp} with numpy."""
    prj = read_prj(prefix + ".prj")
    n = prj["totallength"]
    dt = np.uint64 if prj["integersize"] == 64 else np.uint32
    rd = lambda sfx, d: np.fromfile(prefix + "." + sfx, dtype=d)
    opt = lambda sfx, d: (rd(sfx, d) if os.path.exists(prefix + "." + sfx)
                          else None)
    ssp = opt("ssp", dt)
    qsep, hasq = 0, False
    if prj.get("numofquerysequences", 0) > 0:
        # database and query files are separated at the separator in front
        # of the first query sequence (getqueryseppos)
        hasq = True
        qsep = int(ssp[prj["numofdbsequences"] - 1])
    return Index(n, prj["prefixlength"], numofchars, rd("tis", np.uint8),
                 rd("suf", dt), rd("lcp", np.uint8), rd("llv", dt),
                 rd("bck", dt), opt("bwt", np.uint8), opt("sti1", np.uint8),
                 ssp, prj["numofsequences"], qsep, hasq, prj)


def recommended_prefixlength(numofchars, n):
    return int(oracle_lib().orc_recommendedprefixlength(numofchars, n))


def oracle_build_index(tis, numofchars=4, prefixlength=None, ssp=None,
                       numofsequences=None, bits=64, querysepposition=0,
                       hasqueries=False):
    """CPU construction of the mkvtree tables (oracle/vsindex.c)."""
    tis = np.ascontiguousarray(tis, dtype=np.uint8)
    n = tis.shape[0]
    pl = (recommended_prefixlength(numofchars, n) if prefixlength is None
          else prefixlength)
    suf = np.zeros(n + 1, np.uint64)
    lcp = np.zeros(n + 1, np.uint8)
    llv = np.zeros(2 * (n + 1), np.uint64)
    bck = np.zeros(2 * numofchars ** pl, np.uint64)
    bwt = np.zeros(n + 1, np.uint8)
    sti1 = np.zeros(n + 1, np.uint8)
    nllv = oracle_lib().orc_build_tables(
        tis.ctypes.data, n, numofchars, pl, suf.ctypes.data, lcp.ctypes.data,
        llv.ctypes.data, n + 1, bck.ctypes.data, bwt.ctypes.data,
        sti1.ctypes.data)
    assert nllv >= 0
    if ssp is None:
        ssp = np.nonzero(tis == SEPARATOR)[0].astype(np.uint64)
    idx = Index(n, pl, numofchars, tis, suf, lcp, llv[:2 * nllv], bck, bwt,
                sti1, ssp, len(ssp) + 1 if numofsequences is None
                else numofsequences, querysepposition, hasqueries)
    return idx.as_width(32) if bits == 32 else idx


def fasta_text(paths, symmap=None):
    """Concatenate the records of FASTA files the way mkvtree does: mapped
    symbols, one SEPARATOR between consecutive sequences.
    -> (tis, ssp, number of sequences per file)"""
    symmap = dna_map() if symmap is None else symmap
    parts, perfile = [], []
    for p in paths:
        recs = read_fasta(p)
        perfile.append(len(recs))
        for _, s in recs:
            parts.append(symmap[np.frombuffer(s, np.uint8)])
    buf = []
    for i, s in enumerate(parts):
        if i:
            buf.append(np.array([SEPARATOR], np.uint8))
        buf.append(s)
    tis = np.concatenate(buf)
    return tis, np.nonzero(tis == SEPARATOR)[0].astype(np.uint64), perfile


# --------------------------------------------------------------------------
# the reference programs (oracle/_ref), when built
# --------------------------------------------------------------------------

def run_mkvtree_ref(args, cwd):
    subprocess.check_call([MKVTREE_REF] + list(args), cwd=cwd,
                          stdout=subprocess.DEVNULL)


def run_vmatch_ref(args, cwd, env=None):
    """-> list of output lines without the '#' comment lines."""
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([VMATCH_REF] + list(args), cwd=cwd, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = [l for l in p.stdout.decode().splitlines()
             if l and not l.startswith("#")]
    return p.returncode, lines, p.stderr.decode()


def parse_vmatch_lines(lines, approx=False):
    """default vmatch columns (Vmatch/echomatch.c:878-1020):
    len1 seq1 rel1 D|P len2 seq2 rel2 dist evalue score identity
    -> structured array (length, dbseq, dbrel, queryseq, querystart);
    approx: -complete -e/-h output, querystart = |dist| (rel2 is 0)"""
    dt = np.dtype([("length", "<u8"), ("dbseq", "<u8"), ("dbrel", "<u8"),
                   ("queryseq", "<u8"), ("querystart", "<u8")])
    out = np.zeros(len(lines), dt)
    for i, l in enumerate(lines):
        f = l.split()
        if approx:
            assert int(f[6]) == 0
            out[i] = (int(f[0]), int(f[1]), int(f[2]), int(f[5]),
                      abs(int(f[7])))
            continue
        out[i] = (int(f[0]), int(f[1]), int(f[2]), int(f[5]), int(f[6]))
        assert int(f[4]) == int(f[0]) and int(f[7]) == 0
    return out


def matches_as_ref(index, m):
    """oracle/product matches -> the tuple layout of parse_vmatch_lines"""
    dt = np.dtype([("length", "<u8"), ("dbseq", "<u8"), ("dbrel", "<u8"),
                   ("queryseq", "<u8"), ("querystart", "<u8")])
    out = np.zeros(m.shape[0], dt)
    seq, rel = index.seq_rel(m["dbstart"])
    out["length"], out["dbseq"], out["dbrel"] = m["length"], seq, rel
    out["queryseq"], out["querystart"] = m["queryseq"], m["querystart"]
    return out


def sorted_matches(m):
    return np.sort(m, order=list(m.dtype.names))


# --------------------------------------------------------------------------
# golden cases (tests/golden, written by scripts/make_golden.py)
# --------------------------------------------------------------------------

import gzip as _gzip
import hashlib as _hashlib
import json as _json
import tempfile as _tempfile

_manifest = None
_expected = None
_cases = {}


def manifest():
    global _manifest
    if _manifest is None:
        with open(os.path.join(GOLDEN, "manifest.json")) as f:
            _manifest = _json.load(f)
    return _manifest


def expected(case, key):
    global _expected
    if _expected is None:
        _expected = np.load(os.path.join(GOLDEN, "expected.npz"))
    return _expected["%s__%s" % (case, key)]


def _golden_fasta(name):
    """path of a (possibly gzipped) golden FASTA, unpacked to a temp file"""
    p = os.path.join(GOLDEN, name)
    if not name.endswith(".gz"):
        return p
    out = os.path.join(_tempfile.gettempdir(),
                       "vsa_golden_%d_%s" % (os.getuid(), name[:-3]))
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(p):
        with _gzip.open(p, "rb") as f, open(out + ".tmp", "wb") as g:
            g.write(f.read())
        os.replace(out + ".tmp", out)
    return out


def synth_c1():
    """(genome codes, query codes, n, nq, m) of golden case c1"""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import vstree_amd as V
    s = manifest()["c1"]["synthetic"]
    g = V.synth_genome(s["n"], s["genome_seed"])
    q = V.synth_queries(g, s["nq"], s["m"], s["query_seed"])
    return g, q, s["n"], s["nq"], s["m"]


def table_md5(index):
    """md5 of the tables in the reference's 64-bit on-disk layout"""
    i64 = index.as_width(64)
    return {"tis": _hashlib.md5(i64.tis.tobytes()).hexdigest(),
            "suf": _hashlib.md5(i64.suf.tobytes()).hexdigest(),
            "lcp": _hashlib.md5(i64.lcp.tobytes()).hexdigest(),
            "llv": _hashlib.md5(i64.llv.tobytes()).hexdigest(),
            "bck": _hashlib.md5(i64.bck.tobytes()).hexdigest(),
            "bwt": _hashlib.md5(i64.bwt.tobytes()).hexdigest(),
            "sti1": _hashlib.md5(i64.sti1.tobytes()).hexdigest()}


def load_case(case):
    """-> (Index built by the CPU oracle builder, Queries or None).
    The tables are checked against the md5 sums of what the reference's
    mkvtree wrote, so every user of a case starts from reference tables."""
    if case in _cases:
        return _cases[case]
    m = manifest()[case]
    pl = m["index"]["prj"]["prefixlength"]
    queries = None
    if "synthetic" in m:
        g, qb, n, nq, mm = synth_c1()
        idx = oracle_build_index(g, 4, pl)
        queries = Queries.uniform(qb, mm)
    else:
        files = [_golden_fasta(f) for f in m["db"]]
        qfiles = [_golden_fasta(f) for f in m.get("indexedquery", [])]
        tis, ssp, perfile = fasta_text(files + qfiles)
        hasq = bool(qfiles)
        qsep = int(ssp[sum(perfile[:len(files)]) - 1]) if hasq else 0
        idx = oracle_build_index(tis, 4, pl, ssp=ssp, querysepposition=qsep,
                                 hasqueries=hasq)
        idx.numofdbsequences = sum(perfile[:len(files)])
        if "query" in m:
            queries = fasta_queries(_golden_fasta(m["query"]))
    got = table_md5(idx)
    assert got == m["index"]["md5"], (case, got, m["index"]["md5"])
    if not hasattr(idx, "numofdbsequences"):
        idx.numofdbsequences = idx.numofsequences
    _cases[case] = (idx, queries)
    return _cases[case]


def index_as_rc_queries(index):
    """the query set of vmatch -p IDX: every sequence of the index reversed
    and complemented on its own (copymultiseqRC, readmulti.c:93-125)"""
    bounds = np.concatenate(([-1], index.ssp.astype(np.int64), [index.n]))
    sym = index.tis.copy()
    start, length = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        s, l = int(a) + 1, int(b - a - 1)
        seg = index.tis[s:s + l][::-1]
        sym[s:s + l] = np.where(seg >= WILDCARD, seg, 3 - seg)
        start.append(s)
        length.append(l)
    return Queries(sym, start, length)


def palindromic_as_ref(index, m):
    """matches of the reverse complements against the index -> what vmatch
    -p IDX prints: position flipped back to the forward strand, of the two
    mirror images the one with the smaller left position
    (Vmatch/procfinal.c:152-167)"""
    dt = np.dtype([("length", "<u8"), ("dbseq", "<u8"), ("dbrel", "<u8"),
                   ("queryseq", "<u8"), ("querystart", "<u8")])
    bounds = np.concatenate(([-1], index.ssp.astype(np.int64), [index.n]))
    seqlen = (bounds[1:] - bounds[:-1] - 1).astype(np.uint64)
    s1, r1 = index.seq_rel(m["dbstart"])
    q = m["queryseq"].astype(np.int64)
    r2 = seqlen[q] - (m["querystart"] + m["length"])
    keep = ~((s1 > m["queryseq"]) | ((s1 == m["queryseq"]) & (r1 > r2)))
    out = np.zeros(int(keep.sum()), dt)
    out["length"], out["dbseq"], out["dbrel"] = (m["length"][keep], s1[keep],
                                                 r1[keep])
    out["queryseq"], out["querystart"] = m["queryseq"][keep], r2[keep]
    return out


def repeats_as_ref(index, m):
    """self matches (length, start1, start2) of an index without queries ->
    the reference's output tuple"""
    dt = np.dtype([("length", "<u8"), ("dbseq", "<u8"), ("dbrel", "<u8"),
                   ("queryseq", "<u8"), ("querystart", "<u8")])
    out = np.zeros(m.shape[0], dt)
    s1, r1 = index.seq_rel(m["dbstart"])
    s2, r2 = index.seq_rel(m["queryseq"])
    out["length"], out["dbseq"], out["dbrel"] = m["length"], s1, r1
    out["queryseq"], out["querystart"] = s2, r2
    return out


def selfmatches_as_ref(index, m):
    """self-MUM records (length, start1, start2) -> the reference's output
    tuple; the second sequence number counts from the first query sequence"""
    dt = np.dtype([("length", "<u8"), ("dbseq", "<u8"), ("dbrel", "<u8"),
                   ("queryseq", "<u8"), ("querystart", "<u8")])
    out = np.zeros(m.shape[0], dt)
    s1, r1 = index.seq_rel(m["dbstart"])
    s2, r2 = index.seq_rel(m["queryseq"])
    out["length"], out["dbseq"], out["dbrel"] = m["length"], s1, r1
    out["queryseq"] = s2 - np.uint64(index.numofdbsequences)
    out["querystart"] = r2
    return out
