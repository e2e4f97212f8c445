#!/usr/bin/env python3
"""md5 sums of the index files the REFERENCE's mkvtree writes for the two big
texts of tests/helpers.py (BIG_CASES): the 100 Mbp synthetic genome of SURVEY
8d and a 20 Mbp text of 200 sequences with long repeats, a tandem array and
runs of wildcards.  -> tests/golden/bigindex.json (data only: md5 sums, the
numbers of the .prj file).

    python3 scripts/make_golden_big.py          (build container: needs
        oracle/_ref/mkvtree_ref, about 4 GB of scratch space and some minutes)

The 64-bit files are the reference's own (LP64 build).  The md5 sums under
"md5_32" are those of the same tables narrowed to 32-bit integers (suf, bck,
llv, skp; integersize=32 in the .prj) -- what `integersize=32` of vsa_mkvtree
must write (.sds and .ssp hold Uint values, too), derived here from the reference's output, not from ours."""
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers as H  # noqa: E402

FILES = ("prj", "al1", "tis", "ois", "des", "sds", "ssp", "suf", "lcp", "llv",
         "bck", "bwt", "sti1", "skp")
WIDE = ("suf", "bck", "llv", "skp", "sds", "ssp")


def md5file(p, narrow=False):
    h = hashlib.md5()
    if narrow:
        a = np.fromfile(p, np.uint64)
        assert int(a.max(initial=0)) < 2 ** 32
        h.update(a.astype(np.uint32).tobytes())
    else:
        with open(p, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                h.update(chunk)
    return h.hexdigest()


def main():
    if not H.have_ref():
        sys.exit("build the reference first: make -f oracle/Makefile.ref")
    out = {}
    for case in H.BIG_CASES:
        wd = tempfile.mkdtemp(dir=os.environ.get("VSA_SCRATCH", "/tmp"))
        try:
            name = case + ".fna"
            t0 = time.time()
            H.write_fasta_fast(os.path.join(wd, name),
                               H.big_case_records(case))
            t1 = time.time()
            H.run_mkvtree_ref(["-db", name, "-dna", "-pl", "-allout"], wd)
            t2 = time.time()
            prefix = os.path.join(wd, name)
            prj = H.read_prj(prefix + ".prj")
            entry = {
                "fasta_md5": md5file(prefix),
                "prj": {k: v for k, v in prj.items()
                        if k not in ("dbfile", "queryfile")},
                "md5": {t: md5file(prefix + "." + t) for t in FILES
                        if os.path.exists(prefix + "." + t)},
                "md5_32": {t: md5file(prefix + "." + t, narrow=True)
                           for t in WIDE if os.path.exists(prefix + "." + t)},
                "reference_mkvtree_s": round(t2 - t1, 1)}
            text = open(prefix + ".prj").read()
            assert "integersize=64" in text
            entry["md5_32"]["prj"] = hashlib.md5(
                text.replace("integersize=64", "integersize=32").encode()
            ).hexdigest()
            out[case] = entry
            print(case, "fasta %.0f s, mkvtree %.0f s" % (t1 - t0, t2 - t1),
                  entry["prj"], flush=True)
        finally:
            shutil.rmtree(wd, ignore_errors=True)
    with open(os.path.join(ROOT, "tests", "golden", "bigindex.json"),
              "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
