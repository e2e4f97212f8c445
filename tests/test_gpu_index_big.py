"""Index-builder parity at the size SURVEY 8f-1 states: `mkvtree -db F -dna
-pl -allout` of the 100 Mbp synthetic genome and of a 20 Mbp text with 200
sequences, planted 5 kb repeats (lcp values in the thousands), a tandem array
and runs of wildcards -- every file vsa_mkvtree writes must have the md5 of
the file the reference's mkvtree wrote (tests/golden/bigindex.json, made by
scripts/make_golden_big.py from oracle/_ref/mkvtree_ref), with 64-bit tables
like the reference's LP64 build and with 32-bit tables (the reference's
tables narrowed).  Contract: Mkvtree/bese.c:27-49,533-590 (suffix order,
lcp / llv), Mkvtree/mkvprocess.c:251-327,583-612 (bck, sti1)."""
import hashlib
import json
import os

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
WIDE = ("suf", "bck", "llv", "skp", "sds", "ssp")


def md5file(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(H.GOLDEN, "bigindex.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", H.BIG_CASES)
def test_gpu_mkvtree_writes_the_references_files_at_size(V, golden, case,
                                                         tmp_path,
                                                         monkeypatch):
    wd = str(tmp_path)
    name = case + ".fna"
    monkeypatch.chdir(wd)            # the .prj records file names as given
    H.write_fasta_fast(name, H.big_case_records(case))
    g = golden[case]
    assert md5file(name) == g["fasta_md5"]   # the very text the reference saw
    V.mkvtree([name], name, integersize=64, withskp=True)
    got = {t: md5file(name + "." + t) for t in g["md5"]}
    assert got == g["md5"]
    prj = H.read_prj(name + ".prj")
    for k, v in g["prj"].items():
        assert prj[k] == v, k
    for t in g["md5"]:
        os.unlink(name + "." + t)
    # 32-bit tables: the reference's, narrowed
    V.mkvtree([name], name, integersize=32, withskp=False)
    want = dict(g["md5"])
    want.update(g["md5_32"])
    want.pop("skp", None)
    got = {t: md5file(name + "." + t) for t in want}
    assert got == want
