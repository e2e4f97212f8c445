"""The host match sink (vstree_amd/csrc/match_sink.c) against what vmatch
itself printed: the golden match lists are turned back into engine records,
formatted, and the md5 of the lines must be the md5 the manifest recorded from
the reference's stdout -- sequence numbers, relative positions, E-values,
scores, identities and column widths included.  No GPU involved."""
import hashlib

import numpy as np
import pytest

import helpers as H
import vstree_amd as V

M = H.manifest()

CASES = [(c, k) for c in sorted(M) for k in sorted(M[c]["runs"])
         if "strands" not in M[c]["runs"][k] and not k.endswith("_short")]


def records(idx, exp, q=None):
    """(length, dbseq, dbrel, queryseq, querystart) -> engine records"""
    starts = np.concatenate(([0], idx.ssp + 1)).astype(np.uint64)
    m = np.zeros(len(exp), V.MATCH_DTYPE)
    m["length"] = exp["length"]
    m["dbstart"] = starts[exp["dbseq"].astype(np.int64)] + exp["dbrel"]
    m["queryseq"], m["querystart"] = exp["queryseq"], exp["querystart"]
    return m


def query_sink(idx, q, kind, **kw):
    total = int(q.length.sum()) + q.nq - 1
    return V.Sink(kind, idx.n, idx.ssp, 4, q.start, q.length, total, **kw)


def kind_of(key):
    if key.startswith("approx_e"):
        return V.SINK_APPROX_EDIST, 0
    if key.startswith("approx_h"):
        return V.SINK_APPROX_HAMMING, 0
    if key.startswith("complete"):
        return V.SINK_COMPLETE, 0
    name = key.partition("_sp")[0]
    digits = "".join(ch for ch in name if ch.isdigit())
    return V.SINK_QUERY, int(digits)


@pytest.mark.parametrize("case,key", CASES)
def test_sink_prints_what_vmatch_printed(case, key):
    idx, q = H.load_case(case)
    run = M[case]["runs"][key]
    exp = H.expected(case, key)
    if key.startswith("selfmum") or (key.startswith("repeats") and
                                     idx.hasqueries):
        # records are (length, start1, start2): rebuild start2 from the
        # query-side sequence number the reference printed
        prj = M[case]["index"]["prj"]
        starts = np.concatenate(([0], idx.ssp + 1)).astype(np.uint64)
        m = np.zeros(len(exp), V.MATCH_DTYPE)
        m["length"] = exp["length"]
        m["dbstart"] = starts[exp["dbseq"].astype(np.int64)] + exp["dbrel"]
        m["queryseq"] = starts[(exp["queryseq"] + np.uint64(
            idx.numofdbsequences)).astype(np.int64)] + exp["querystart"]
        sink = V.Sink(V.SINK_SELF, idx.n, idx.ssp, 4,
                      numofquerysequences=prj["numofquerysequences"],
                      totalquerylength=idx.n - idx.querysepposition - 1,
                      leastlength=int("".join(
                          ch for ch in key if ch.isdigit())))
    elif key.startswith("palindromic"):
        # printed positions are on the forward strand: back to the offset in
        # the reverse complement the engine reported
        rq = H.index_as_rc_queries(idx)
        m = records(idx, exp)
        m["querystart"] = rq.length[exp["queryseq"].astype(np.int64)] - (
            exp["querystart"] + exp["length"])
        sink = V.Sink(V.SINK_QUERY, idx.n, idx.ssp, 4, rq.start, rq.length,
                      idx.n, leastlength=30, palindromic=True,
                      selfpalindromic=True)
    elif key.startswith(("supermax", "repeats", "tandem")):
        starts = np.concatenate(([0], idx.ssp + 1)).astype(np.uint64)
        m = records(idx, exp)
        m["queryseq"] = starts[exp["queryseq"].astype(np.int64)] + \
            exp["querystart"]
        m["querystart"] = 0
        sink = V.Sink(V.SINK_SELF, idx.n, idx.ssp, 4,
                      leastlength=int("".join(
                          ch for ch in key if ch.isdigit())))
    else:
        kind, least = kind_of(key)
        m = records(idx, exp)
        sink = query_sink(idx, q, kind, leastlength=least)
    text = sink.format(m)
    assert text.count(b"\n") == run["lines"]
    assert hashlib.md5(text).hexdigest() == run["md5_lines"], \
        text[:300].decode()


def rc_queries(q):
    sym = q.symbols.copy()
    for s, l in zip(q.start, q.length):
        s, l = int(s), int(l)
        seg = q.symbols[s:s + l][::-1]
        sym[s:s + l] = np.where(seg == H.WILDCARD, H.WILDCARD, 3 - seg)
    return H.Queries(sym, q.start, q.length)


@pytest.mark.parametrize("key", ["complete_dp", "mum20_dp"])
def test_both_strands(key):
    """vmatch -d -p: the direct pass, then the pass over the reverse
    complements with flag P and the position flipped back to the forward
    strand (procfinal.c:152-158); match lists from the CPU oracle"""
    idx, q = H.load_case("c1")
    run = M["c1"]["runs"][key]
    out = b""
    for pal, qq in ((False, q), (True, rc_queries(q))):
        if key.startswith("complete"):
            m, kind, least = H.oracle_complete(idx, qq), V.SINK_COMPLETE, 0
        else:
            m = H.oracle_querymatches(idx, qq, 20, mum=True, speedup=2)
            kind, least = V.SINK_QUERY, 20
        out += query_sink(idx, qq, kind, leastlength=least,
                          palindromic=pal).format(m)
    assert out.count(b"\n") == run["lines"]
    assert hashlib.md5(out).hexdigest() == run["md5_lines"]


def test_options_and_errors():
    idx, q = H.load_case("micro")
    m = records(idx, H.expected("micro", "complete"))
    full = query_sink(idx, q, V.SINK_COMPLETE).format(m).split(b"\n")[0]
    bare = query_sink(idx, q, V.SINK_COMPLETE, showmode=(
        V.SHOW_NODIST | V.SHOW_NOEVALUE | V.SHOW_NOSCORE |
        V.SHOW_NOIDENTITY)).format(m).split(b"\n")[0]
    assert full.startswith(bare) and len(bare.split()) == 7
    # matchokay: matches below the least length are dropped
    assert query_sink(idx, q, V.SINK_COMPLETE,
                      leastlength=1000).format(m) == b""
    bad = m.copy()
    bad["queryseq"] = 999
    with pytest.raises(V.VsaError):
        query_sink(idx, q, V.SINK_COMPLETE).format(bad)
