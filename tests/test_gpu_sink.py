"""Engine -> host sink, without any reference program in the loop: the GPU's
match lists formatted by vstree_amd/csrc/match_sink.c must have the md5 of the
lines the reference vmatch printed (tests/golden/manifest.json)."""
import hashlib

import numpy as np
import pytest

import helpers as H
from test_gpu_parity import gpu_index, gpu_queries
from test_sink import query_sink

pytestmark = pytest.mark.gpu
M = H.manifest()


@pytest.mark.parametrize("case,key", [
    ("c1", "complete"), ("c1", "mum20"), ("c1", "mumcand20"),
    ("c1", "mem20_sp0"), ("c1", "approx_e2"), ("c1", "approx_h2"),
    ("c5", "approx_e2"), ("c5", "approx_e3"), ("largepat", "complete"),
    ("grumbach", "mum14"), ("micro", "mem3_sp0")])
def test_gpu_matches_through_the_sink_equal_vmatch_stdout(V, case, key):
    idx, q = H.load_case(case)
    gi, gq = gpu_index(V, case), gpu_queries(V, q)
    run = M[case]["runs"][key]
    least = 0
    if key.startswith("approx_"):
        doedist = key[7] == "e"
        m = V.findapproxcompletematches(gi, gq, doedist, int(key[8:])).fetch()
        kind = V.SINK_APPROX_EDIST if doedist else V.SINK_APPROX_HAMMING
    elif key.startswith("complete"):
        m, kind = V.findcompletematches(gi, gq).fetch(), V.SINK_COMPLETE
    else:
        name = key.partition("_sp")[0]
        least = int("".join(ch for ch in name if ch.isdigit()))
        m = V.findquerymatches(gi, gq, least, mum=name.startswith("mum"),
                               cand="cand" in name,
                               speedup=0 if key.endswith("_sp0") else 2
                               ).fetch()
        kind = V.SINK_QUERY
    text = query_sink(idx, q, kind, leastlength=least).format(m)
    assert text.count(b"\n") == run["lines"]
    assert hashlib.md5(text).hexdigest() == run["md5_lines"]


def test_self_index_mums_through_the_sink(V):
    idx, _ = H.load_case("grumbach_all")
    gi = gpu_index(V, "grumbach_all")
    run = M["grumbach_all"]["runs"]["selfmum14"]
    m = V.findmaximaluniquematches(gi, 14).fetch()
    sink = V.Sink(V.SINK_SELF, idx.n, idx.ssp, 4, numofquerysequences=1,
                  totalquerylength=idx.n - idx.querysepposition - 1,
                  leastlength=14)
    text = sink.format(m)
    assert hashlib.md5(text).hexdigest() == run["md5_lines"]
