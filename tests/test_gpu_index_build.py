"""The GPU index builder against the reference's tables: md5 sums of what
mkvtree wrote (golden manifest) and the CPU table oracle on seeded inputs."""
import hashlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
M = H.manifest()


def md5s(t):
    return {k: hashlib.md5(np.ascontiguousarray(t[k]).astype(
        np.uint64 if k in ("suf", "llv", "bck") else np.uint8).tobytes()
    ).hexdigest() for k in ("tis", "suf", "lcp", "llv", "bck", "bwt")}


@pytest.mark.parametrize("case", sorted(M))
def test_built_tables_have_the_references_md5(V, case):
    idx, _ = H.load_case(case)
    gi = V.Index.build(idx.tis, 4, idx.prefixlength)
    got = md5s(gi.download())
    want = {k: M[case]["index"]["md5"][k] for k in got}
    assert got == want
    info = gi.info()
    assert info.largelcpvalues == M[case]["index"]["prj"]["largelcpvalues"]


def test_recommended_prefixlength_is_the_references(V):
    for case in sorted(M):
        idx, _ = H.load_case(case)
        gi = V.Index.build(idx.tis, 4, 0)
        assert gi.info().prefixlength == M[case]["index"]["prj"][
            "prefixlength"]


def check_against_oracle(V, tis, pl=None, numofchars=4):
    want = H.oracle_build_index(tis, numofchars, pl)
    gi = V.Index.build(tis, numofchars, want.prefixlength)
    got = gi.download()
    for k in ("suf", "lcp", "llv", "bck", "bwt"):
        assert np.array_equal(got[k].astype(np.uint64),
                              getattr(want, k).astype(np.uint64)), k
    return gi, want


def test_random_text_with_specials(V):
    rng = np.random.default_rng(99)
    tis = rng.integers(0, 4, size=200000).astype(np.uint8)
    tis[rng.integers(0, len(tis), size=300)] = H.WILDCARD
    tis[rng.integers(0, len(tis), size=40)] = H.SEPARATOR
    tis[1000:1040] = H.WILDCARD          # a run of wildcards
    tis[-3:] = H.WILDCARD                # special suffix
    tis[0] = H.WILDCARD                  # special prefix
    check_against_oracle(V, tis)


def test_repetitive_text_needs_many_doubling_rounds(V):
    rng = np.random.default_rng(3)
    unit = rng.integers(0, 4, size=977).astype(np.uint8)
    tis = np.concatenate([np.tile(unit, 20), np.zeros(5000, np.uint8),
                          np.tile(np.array([0, 1], np.uint8), 3000),
                          rng.integers(0, 4, 3000).astype(np.uint8),
                          np.tile(unit, 3)])
    gi, want = check_against_oracle(V, tis, 5)
    assert want.nllv > 1000


def test_tiny_and_degenerate_texts(V):
    for tis in ([0], [0, 0], [3, 2, 1, 0], [254], [255, 255], [0, 254, 0],
                [1] * 70, list(range(4)) * 9):
        check_against_oracle(V, np.array(tis, np.uint8), 1)


def test_protein_like_alphabet(V):
    rng = np.random.default_rng(11)
    tis = rng.integers(0, 20, size=30000).astype(np.uint8)
    tis[rng.integers(0, len(tis), size=30)] = H.WILDCARD
    tis[7000] = H.SEPARATOR
    gi, want = check_against_oracle(V, tis, 2, numofchars=20)
    # and the search kernels run on it
    seqs = [tis[p:p + 12] for p in rng.integers(0, 29000, size=200)]
    q = H.Queries.from_list(seqs)
    gq = V.Queries.from_host(q.symbols, q.start, q.length)
    assert np.array_equal(V.findcompletematches(gi, gq).fetch(),
                          H.oracle_complete(want, q))
    for sp in (0, 2):
        assert np.array_equal(
            V.findquerymatches(gi, gq, 4, speedup=sp).fetch(),
            H.oracle_querymatches(want, q, 4, speedup=sp)), sp


def test_search_on_built_index_equals_golden(V):
    """end to end: text -> GPU index -> GPU search == reference output"""
    idx, q = H.load_case("c1")
    gi = V.Index.build(idx.tis, 4, 0)
    gq = V.Queries.from_host(q.symbols, q.start, q.length)
    got = H.matches_as_ref(idx, V.findcompletematches(gi, gq).fetch())
    assert np.array_equal(got, H.expected("c1", "complete"))
    got = H.matches_as_ref(idx, V.findquerymatches(gi, gq, 20,
                                                   mum=True).fetch())
    assert np.array_equal(got, H.expected("c1", "mum20"))


# ---- wide tables (n + 1 >= 2^32 in production; forced here) -----------------

def test_wide_builder_writes_the_same_tables(V, monkeypatch):
    """VSA_FORCE_WIDE=1: the 64-bit instantiation of the builder (prefix
    doubling on 80-bit composite keys) and of the deep-table kernels, on the
    golden texts and on the seeded ones above"""
    monkeypatch.setenv("VSA_FORCE_WIDE", "1")
    for case in ("c1", "grumbach", "wildcards"):
        idx, _ = H.load_case(case)
        gi = V.Index.build(idx.tis, 4, idx.prefixlength)
        info = gi.info()
        assert info.device_integersize == 64 and info.deepprefix > 0
        got = md5s(gi.download())
        assert got == {k: M[case]["index"]["md5"][k] for k in got}, case
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 4, size=977).astype(np.uint8)
    tis = np.concatenate([np.tile(unit, 20), np.zeros(5000, np.uint8),
                          np.tile(np.array([0, 1], np.uint8), 3000),
                          rng.integers(0, 4, 3000).astype(np.uint8),
                          np.tile(unit, 3)])
    tis[rng.integers(0, len(tis), size=30)] = H.WILDCARD
    tis[12000] = H.SEPARATOR
    gi, want = check_against_oracle(V, tis, 5)
    assert gi.info().device_integersize == 64 and want.nllv > 1000
    for t in ([0], [3, 2, 1, 0], [255, 255], [1] * 70):
        check_against_oracle(V, np.array(t, np.uint8), 1)
    # sti1 from wide tables
    idx, _ = H.load_case("grumbach")
    gi = V.Index.build(idx.tis, 4, idx.prefixlength)
    assert np.array_equal(gi.make_sti1(), H.sti1_from_tables(
        idx.suf, idx.lcp, idx.prefixlength))


def test_search_on_a_wide_built_index(V, monkeypatch):
    """text -> wide GPU index (deep tables included) -> every query mode"""
    monkeypatch.setenv("VSA_FORCE_WIDE", "1")
    idx, q = H.load_case("c1")
    gi = V.Index.build(idx.tis, 4, 0)
    assert gi.info().device_integersize == 64 and gi.info().deepprefix > 0
    gq = V.Queries.from_host(q.symbols, q.start, q.length)
    got = H.matches_as_ref(idx, V.findcompletematches(gi, gq).fetch())
    assert np.array_equal(got, H.expected("c1", "complete"))
    for key, kw in (("mum20", dict(mum=True)),
                    ("mumcand20", dict(mum=True, cand=True)),
                    ("mem20_sp0", dict(speedup=0)), ("mem20_sp2", {})):
        if key not in M["c1"]["runs"]:
            continue
        got = H.matches_as_ref(idx, V.findquerymatches(gi, gq, 20,
                                                       **kw).fetch())
        assert np.array_equal(got, H.expected("c1", key)), key
