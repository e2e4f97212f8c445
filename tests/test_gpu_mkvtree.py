"""vsa_mkvtree = `mkvtree -db .. [-q ..] -dna -pl -allout` on the GPU: every
file it writes must have the md5 of the file the reference's mkvtree wrote
(golden manifest), and the UNMODIFIED reference vmatch must produce its golden
output from the GPU-built index."""
import hashlib
import os

import numpy as np
import pytest

import helpers as H
from test_gpu_dropin import stage, MKV

pytestmark = pytest.mark.gpu
M = H.manifest()


def md5file(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


@pytest.mark.parametrize("case", sorted(MKV))
def test_gpu_mkvtree_writes_the_references_files(V, case, tmp_path,
                                                 monkeypatch):
    wd = str(tmp_path)
    stage(case, wd)
    monkeypatch.chdir(wd)          # the .prj records file names as given
    args = M[case]["index"]["mkvargs"]
    name = M[case]["index"]["indexname"]
    db = [args[i + 1] for i, a in enumerate(args) if a == "-db"]
    qf = [args[i + 1] for i, a in enumerate(args) if a == "-q"]
    V.mkvtree(db, name, qf)
    want = M[case]["index"]["md5_allfiles"]
    got = {t: md5file(name + "." + t) for t in want}
    assert got == want
    # nothing the reference did not write
    mine = {f.split(".")[-1] for f in os.listdir(wd)
            if f.startswith(name + ".")}
    if "-allout" not in args:
        mine.discard("skp")    # asked for table by table, skp not among them
    assert mine == set(want)


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
def test_reference_vmatch_runs_on_gpu_built_index(V, tmp_path, monkeypatch):
    wd = str(tmp_path)
    stage("grumbach", wd)
    monkeypatch.chdir(wd)
    V.mkvtree(["humhbb.fna"], "humhbb.fna")
    for key in ("mem14_sp2", "mum14", "mumcand14"):
        run = M["grumbach"]["runs"][key]
        rc, lines, err = H.run_vmatch_ref(run["args"], wd)
        assert rc == 0, err
        assert hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest() \
            == run["md5_lines"], key


def test_gpu_mkvtree_32bit_files_and_errors(V, tmp_path, monkeypatch):
    wd = str(tmp_path)
    stage("micro", wd)
    monkeypatch.chdir(wd)
    V.mkvtree(["db.fna"], "i32", integersize=32, withskp=False)
    prj = H.read_prj("i32.prj")
    assert prj["integersize"] == 32
    assert not os.path.exists("i32.skp")
    idx = H.load_mkvtree_index("i32")
    ref, _ = H.load_case("micro")
    for t in ("suf", "lcp", "bck", "bwt", "sti1", "llv"):
        assert np.array_equal(getattr(idx, t).astype(np.uint64),
                              getattr(ref, t).astype(np.uint64)), t
    gi = V.Index.open("i32")          # and the C reader takes it back
    assert gi.info().totallength == ref.n
    with open("bad.fna", "w") as f:
        f.write(">x\nacgtxacgt\n")
    with pytest.raises(V.VsaError) as e:
        V.mkvtree(["bad.fna"], "bad")
    assert "Illegal character 'x' in file \"bad.fna\" line 2" in \
        e.value.message
    with pytest.raises(V.VsaError) as e:
        V.mkvtree(["nothing.fna"], "bad")
    assert "cannot open file" in e.value.message
