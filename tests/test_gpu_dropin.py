"""Drop-in check: the reference's own vmatch program (CLI, processfinal sink,
E-values, output formatting) with the three Vmengine entry points of this path
wrapped onto the GPU library (integration/vmengine_shim.c) must print what the
unmodified reference printed (md5 of the output lines in the golden manifest).

Needs the prebuilt binaries oracle/_ref/mkvtree_ref (writes the index files)
and integration/_build/vmatch_gpu; both are built in the build container and
travel to the GPU box."""
import gzip
import hashlib
import os
import shutil
import subprocess

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
M = H.manifest()
VMATCH_GPU = os.path.join(H.ROOT, "integration", "_build", "vmatch_gpu")

needs_binaries = pytest.mark.skipif(
    not (os.access(VMATCH_GPU, os.X_OK) and os.access(H.MKVTREE_REF, os.X_OK)),
    reason="integration/_build/vmatch_gpu or oracle/_ref/mkvtree_ref not built")


def stage(case, wd):
    """input files of a golden case under the names the manifest's command
    lines use"""
    m = M[case]
    if "synthetic" in m:
        g, qb, n, nq, mm = H.synth_c1()
        H.write_fasta(wd + "/genome.fna", [("synthetic_genome seed=42", g)])
        H.write_fasta(wd + "/queries.fna",
                      [("q%d" % i, qb[i * mm:(i + 1) * mm])
                       for i in range(nq)], width=1000)
        return
    for name in m.get("db", []) + m.get("indexedquery", []) + (
            [m["query"]] if "query" in m else []):
        src = os.path.join(H.GOLDEN, name)
        dst = name[:-3] if name.endswith(".gz") else name
        for prefix in ("micro_", "c5_", "c6_"):
            dst = dst[len(prefix):] if dst.startswith(prefix) else dst
        if name.endswith(".gz"):
            with gzip.open(src, "rb") as f, open(wd + "/" + dst, "wb") as g:
                g.write(f.read())
        elif not os.path.exists(wd + "/" + dst):
            shutil.copy(src, wd + "/" + dst)
    if case == "grumbach":
        shutil.copy(os.path.join(H.GOLDEN, "short.fna"), wd + "/short.fna")


MKV = {"largepat": ["-db", "ychrIII.fna"], "micro": ["-db", "db.fna"],
       "wildcards": ["-db", "Wildcards.fna"],
       "grumbach": ["-db", "humhbb.fna"],
       "grumbach_all": ["-indexname", "all", "-db", "humhbb.fna", "-q",
                        "humdystrop.fna"],
       "c1": ["-db", "genome.fna"], "c5": ["-db", "db.fna"],
       "c6": ["-db", "db.fna"],
       "at1mb": ["-indexname", "atindex", "-db", "at1MB"]}


def run_gpu_vmatch(args, wd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([VMATCH_GPU] + list(args), cwd=wd, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = [l for l in p.stdout.decode().splitlines()
             if l and not l.startswith("#")]
    return p.returncode, lines, p.stderr.decode()


@needs_binaries
@pytest.mark.parametrize("case", sorted(MKV))
def test_vmatch_with_gpu_engine_prints_reference_output(case, tmp_path):
    wd = str(tmp_path)
    stage(case, wd)
    H.run_mkvtree_ref(MKV[case] + ["-dna", "-pl", "-allout"], wd)
    for key, run in sorted(M[case]["runs"].items()):
        rc, lines, err = run_gpu_vmatch(run["args"], wd,
                                        {"VMATCH_GPU_TRACE": "1"})
        assert (rc != 0) == (run["rc"] != 0), (key, err)
        # the engine call really ran on the GPU
        assert "on the GPU" in err, (case, key, err)
        if run["rc"] != 0:
            # same message as the reference, after the same matches
            assert run["stderr"].split(": ", 1)[1] in err
        # byte for byte, the default algorithm 2 (-qspeedup 2) included
        md5 = hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest()
        assert md5 == run["md5_lines"], (case, key)


@needs_binaries
@pytest.mark.parametrize("case", ["c1", "c6", "grumbach", "micro"])
def test_vmatch_on_several_replicas_prints_reference_output(case, tmp_path):
    """VMATCH_GPU_DEVICES=0,0,0: the shim's multi-GPU path (vsa_multi_*, one
    host thread per replica) with three replicas on the one GPU of the box --
    same stdout as the reference, engine calls traced as 'all replicas'."""
    wd = str(tmp_path)
    stage(case, wd)
    H.run_mkvtree_ref(MKV[case] + ["-dna", "-pl", "-allout"], wd)
    seen = 0
    for key, run in sorted(M[case]["runs"].items()):
        # (approximate matching over replicas since round 3)
        if not key.startswith(("complete", "mem", "mum", "approx")):
            continue
        rc, lines, err = run_gpu_vmatch(
            run["args"], wd, {"VMATCH_GPU_TRACE": "1",
                              "VMATCH_GPU_DEVICES": "0,0,0"})
        assert (rc != 0) == (run["rc"] != 0), (key, err)
        assert "all replicas on the GPU" in err, (case, key, err)
        if run["rc"] != 0:
            assert run["stderr"].split(": ", 1)[1] in err
        md5 = hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest()
        assert md5 == run["md5_lines"], (case, key)
        seen += 1
    assert seen >= 4


@needs_binaries
def test_gpu_switch_off_gives_the_reference_engine(tmp_path):
    wd = str(tmp_path)
    stage("micro", wd)
    H.run_mkvtree_ref(MKV["micro"] + ["-dna", "-pl", "-allout"], wd)
    run = M["micro"]["runs"]["mem3_sp2"]
    rc, lines, err = run_gpu_vmatch(run["args"], wd, {"VMATCH_GPU": "0"})
    assert rc == 0
    assert hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest() == \
        run["md5_lines"]


@needs_binaries
def test_selection_function_plugin_still_sees_every_match(tmp_path):
    """SelectBundle (include/select.h:35-51) is called from the reference's
    processfinal, which the GPU engine feeds: -selfun with a tiny plugin that
    keeps matches with an even database position."""
    wd = str(tmp_path)
    stage("grumbach", wd)
    H.run_mkvtree_ref(MKV["grumbach"] + ["-dna", "-pl", "-allout"], wd)
    src = wd + "/seleven.c"
    with open(src, "w") as f:
        f.write('typedef struct { unsigned long idnumber, Storeflag, '
                'Storedistance, Storeposition1, Storelength1, Storeposition2,'
                ' Storelength2, Storeseqnum1, Storerelpos1, Storeseqnum2, '
                'Storerelpos2; double StoreEvalue; } SM;\n'
                'long selectmatch(void *a, void *d, void *q, SM *m)'
                '{ return (m->Storeposition1 % 2 == 0) ? 1 : 0; }\n')
    so = wd + "/seleven.so"
    try:
        subprocess.check_call(["gcc", "-shared", "-fPIC", src, "-o", so])
    except Exception:
        pytest.skip("no compiler on this box")
    args = ["-l", "14", "-selfun", so, "-q", "humdystrop.fna", "humhbb.fna"]
    rc, lines, err = run_gpu_vmatch(args, wd)
    assert rc == 0, err
    rc0, lines0, err0 = run_gpu_vmatch(args, wd, {"VMATCH_GPU": "0"})
    assert rc0 == 0, err0
    assert sorted(lines) == sorted(lines0)
    assert 0 < len(lines) < M["grumbach"]["runs"]["mem14_sp2"]["lines"]


# ---- boundary B2: the reference's dlopen hook for complete matches ---------

PLUGIN_DIR = os.path.join(H.ROOT, "integration", "_build")
PLUGIN = "cpridxps_amd.so"

needs_plugin = pytest.mark.skipif(
    not (os.path.exists(os.path.join(PLUGIN_DIR, PLUGIN)) and
         os.access(H.VMATCH_REF, os.X_OK) and
         os.access(H.MKVTREE_REF, os.X_OK)),
    reason="integration/_build/cpridxps_amd.so or oracle/_ref not built")


def run_ref_vmatch_with_plugin(args, wd):
    """the UNMODIFIED reference program; `-complete` gets the plugin's bare
    name as its argument (parsevm.c:1138-1179), found via LD_LIBRARY_PATH"""
    args = list(args)
    k = args.index("-complete")
    args.insert(k + 1, PLUGIN)
    e = dict(os.environ)
    e["LD_LIBRARY_PATH"] = PLUGIN_DIR + os.pathsep + e.get(
        "LD_LIBRARY_PATH", "")
    p = subprocess.run([H.VMATCH_REF] + args, cwd=wd, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = [l for l in p.stdout.decode().splitlines()
             if l and not l.startswith("#")]
    return p.returncode, lines, p.stderr.decode()


@needs_plugin
@pytest.mark.parametrize("case,key", [
    ("c1", "complete"), ("c1", "complete_dp"), ("largepat", "complete"),
    ("micro", "complete"), ("grumbach", "complete_short")])
def test_reference_vmatch_with_cpridxps_plugin(case, key, tmp_path):
    """vmatch -complete cpridxps_amd.so -q Q IDX on the unmodified reference
    binary prints what vmatch -complete -q Q IDX printed: one GPU launch per
    pass behind the per-query vpluginsearch calls"""
    wd = str(tmp_path)
    stage(case, wd)
    H.run_mkvtree_ref(MKV[case] + ["-dna", "-pl", "-allout"], wd)
    run = M[case]["runs"][key]
    rc, lines, err = run_ref_vmatch_with_plugin(run["args"], wd)
    assert (rc != 0) == (run["rc"] != 0), err
    if run["rc"] != 0:
        assert run["stderr"].split(": ", 1)[1] in err
    md5 = hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest()
    assert md5 == run["md5_lines"], (case, key, lines[:3])
