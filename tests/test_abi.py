"""CPU-side checks of the product library: it loads without a GPU, exports
every symbol include/vstree_amd.h declares, its host-only entry points
(synthetic generator, index reader errors) behave, and nothing in the product
tree touches the oracle."""
import ctypes as C
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import helpers as H


def header_symbols():
    text = open(os.path.join(H.ROOT, "include", "vstree_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vsa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(V):
    syms = header_symbols()
    assert len(syms) >= 30
    lib = C.CDLL(V.LIBPATH)
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
    # the variant for processes that bring their own HIP runtime exports the
    # same ABI (it cannot be dlopen'ed here: no runtime is loaded)
    out = subprocess.check_output(["nm", "-D", "--defined-only",
                                   V.LIBPATH_NORT]).decode()
    have = {l.split()[-1] for l in out.splitlines() if l}
    assert set(syms) <= have
    # and the Python binding covers the same set
    assert sorted(V.ABI_SYMBOLS) == syms


def test_multi_gpu_library_exports_its_header(V):
    """include/vstree_amd_multi.h <-> libvstree_amd_multi.so (symbol table
    only: loading it needs RCCL and a HIP runtime with a device)"""
    text = open(os.path.join(H.ROOT, "include", "vstree_amd_multi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(vsa_multi_[a-z0-9_]+)\s*\(", text)))
    assert len(syms) >= 9
    path = os.path.join(os.path.dirname(V.LIBPATH), "libvstree_amd_multi.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only",
                                   path]).decode()
    have = {l.split()[-1] for l in out.splitlines() if l}
    assert set(syms) <= have
    # no kernels of its own, and it needs the product library
    need = subprocess.check_output(["readelf", "-d", path]).decode()
    assert "libvstree_amd.so" in need and "librccl" in need


def test_which_hip_runtime_the_mirror_binds_to_can_be_said_explicitly():
    """VSTREE_AMD_RUNTIME=own|host|auto (vstree_amd/__init__.py): the explicit
    values do not depend on the order of imports and fail where they cannot
    hold; the multi-GPU mirror refuses the library variant it was not linked
    with"""
    import sys

    def run(value, code):
        c = subprocess.run([sys.executable, "-c", code],
                           env=dict(os.environ, VSTREE_AMD_RUNTIME=value),
                           cwd=H.ROOT, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE)
        return c.returncode, c.stdout.decode(), c.stderr.decode()

    show = "import vstree_amd as V; print(V.LIBPATH)"
    rc, out, err = run("own", show)
    assert rc == 0 and out.strip().endswith("libvstree_amd.so")
    rc, out, err = run("host", show)
    assert rc != 0 and "no HIP runtime is mapped" in err
    rc, out, err = run("bogus", show)
    assert rc != 0 and "expected own, host or auto" in err
    # torch first: auto and host take the variant without a runtime of its
    # own, own refuses, and so does the multi-GPU mirror
    rc, out, err = run("auto", "import torch; " + show)
    assert rc == 0 and out.strip().endswith("libvstree_amd_nort.so"), err
    rc, out, err = run("own", "import torch; " + show)
    assert rc != 0 and "two HIP runtimes" in err
    rc, out, err = run("auto", "import torch; import vstree_amd.multi")
    assert rc != 0 and "without torch" in err


def test_synthetic_generator_matches_recorded_md5(V):
    m = H.manifest()["c1"]
    g, q, n, nq, mm = H.synth_c1()
    assert hashlib.md5(g.tobytes()).hexdigest() == m["md5_genome_codes"]
    assert hashlib.md5(q.tobytes()).hexdigest() == m["md5_query_codes"]
    # the plan materialises the same queries
    pos, sub, step = V.synth_query_plan(n, nq, mm)
    q2 = np.stack([g[int(p):int(p) + mm] for p in pos]).copy()
    for i in np.nonzero(sub != V.NO_SUBST)[0]:
        q2[i, sub[i]] = (q2[i, sub[i]] + step[i]) % 4
    assert np.array_equal(q2.ravel(), q)
    # splitmix64 is addressable by index
    assert (V.lib.vsa_splitmix64_at(42, 0) >> 62) == g[0]
    assert (V.lib.vsa_splitmix64_at(42, 999) >> 62) == g[999]


def test_pack_reads_is_host_code_and_round_trips(V):
    """vsa_pack_reads (no GPU): rows at two bits per symbol, first symbol in
    the top bits, a flag byte; reads with a special symbol on the side list"""
    rng = np.random.default_rng(3)
    for m in (5, 28, 29, 32, 100, 124, 150):
        nq = 200
        sym = rng.integers(0, 4, nq * m).astype(np.uint8)
        sym[7 * m + m // 2] = V.WILDCARD
        sym[90 * m] = V.WILDCARD
        rows, special, ns = V.pack_reads(sym, nq, m)
        W = int(V.lib.vsa_packed_words(m))
        assert W == (2 * m + 8 + 63) // 64 and ns == 2
        r = rows.reshape(nq, W)
        flagged = np.flatnonzero(r[:, W - 1] & np.uint64(0xFF))
        assert list(flagged) == [7, 90]
        assert np.array_equal(special[:m], sym[7 * m:8 * m])
        assert int(r[90, 0]) >> 8 == 1 and int(r[7, 0]) >> 8 == 0
        for i in (0, 8, 199):
            got = [(int(r[i, j // 32]) >> (62 - 2 * (j % 32))) & 3
                   for j in range(m)]
            assert got == list(sym[i * m:(i + 1) * m])
            # nothing behind the last symbol but the flag byte
            used = 2 * (m - 32 * (W - 1)) if m > 32 * (W - 1) else 0
            assert int(r[i, W - 1]) & ((1 << (64 - used)) - 1) == 0
    # eight symbols per step (PEXT) == one symbol per step, one thread == three
    # (the side list is numbered in the order of the reads either way)
    for m in (5, 29, 100, 150):
        nq = 200000
        sym = rng.integers(0, 4, nq * m).astype(np.uint8)
        for i in (7, 90, 150000, nq - 1):
            sym[i * m + int(rng.integers(0, m))] = V.WILDCARD
        os.environ["VSA_PACK_SCALAR"] = "1"
        try:
            r0, s0, n0 = V.pack_reads(sym, nq, m)
        finally:
            del os.environ["VSA_PACK_SCALAR"]
        W = int(V.lib.vsa_packed_words(m))
        assert n0 == 4 and [int(x) >> 8 for x in
                            r0.reshape(nq, W)[[7, 90, 150000, nq - 1], 0]] == \
            [0, 1, 2, 3]
        for threads in (1, 3):
            r1, s1, n1 = V.pack_reads(sym, nq, m, threads=threads)
            assert n1 == n0 and np.array_equal(r1, r0) and \
                np.array_equal(s1, s0), (m, threads)
    m, nq = 100, 200
    sym = rng.integers(0, 4, nq * m).astype(np.uint8)
    sym[7 * m + 50] = sym[90 * m] = V.WILDCARD
    W = int(V.lib.vsa_packed_words(m))
    # the side list is too small: the reference-style error, nothing silent
    rows = np.zeros(nq * W, np.uint64)
    special = np.zeros(m, np.uint8)
    ns = C.c_uint64(0)
    rc = V.lib.vsa_pack_reads(V._ptr(sym), nq, m, m, V._ptr(rows),
                              V._ptr(special), 1, C.byref(ns))
    assert rc == -2 and "special symbol" in V.messagespace()


def test_index_open_reports_reference_style_errors(V, tmp_path):
    with pytest.raises(V.VsaError) as e:
        V.Index.open(str(tmp_path / "nothing"))
    assert "cannot open" in e.value.message
    # a .prj with an integer size that is neither 32 nor 64
    p = tmp_path / "bad"
    (tmp_path / "bad.prj").write_text(
        "totallength=10\nprefixlength=1\nlargelcpvalues=0\n"
        "integersize=16\nlittleendian=1\n")
    (tmp_path / "bad.al1").write_text("aA\ncC\ngG\ntTuU\nnN\n")
    with pytest.raises(V.VsaError) as e:
        V.Index.open(str(p))
    assert "integer size" in e.value.message
    # table of the wrong size (the reference's EXPECTED check)
    (tmp_path / "bad.prj").write_text(
        "totallength=10\nprefixlength=1\nlargelcpvalues=0\n"
        "integersize=64\nlittleendian=1\n")
    (tmp_path / "bad.tis").write_bytes(b"\0" * 9)
    with pytest.raises(V.VsaError) as e:
        V.Index.open(str(p))
    assert "expected 10" in e.value.message


def test_product_tree_never_touches_the_oracle():
    bad = []
    for base in ("vstree_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(H.ROOT, base)):
            if "_build" in dp or "__pycache__" in dp:
                continue
            for f in fn:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"liboracle|vsoracle|orc_[a-z]+\(|oracle/", text):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
