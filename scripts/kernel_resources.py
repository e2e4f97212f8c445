#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in a HIP object or library.

    make -C vstree_amd/csrc COMPRESS= OBJDIR=/tmp/vsa_plain /tmp/vsa_plain/esa_search.o
    python scripts/kernel_resources.py /tmp/vsa_plain/esa_search.o [pattern]

(the library's own objects are built with --offload-compress; the variant of
the compressed bundle hipcc 7.2 writes is not one this script unpacks)

Finds the clang offload bundles in the file, takes the gfx950 code objects out
and reads their metadata notes with llvm-readelf.  (Bundles may be compressed
-- "CCOB" -- in which case clang-offload-bundler does the unpacking.)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    at = 0
    while True:
        at = blob.find(MAGIC, at)
        if at < 0:
            return
        n, = struct.unpack_from("<Q", blob, at + 24)
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        at += 24


def main():
    path = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    with open(path, "rb") as f:
        blob = f.read()
    rows = []
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as t:
            t.write(co)
        txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", t.name],
                             stdout=subprocess.PIPE).stdout.decode()
        os.unlink(t.name)
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "name" and v.startswith("_Z") or k == "name" and "k_" in v:
                cur["name"] = v
            if k in ("vgpr_count", "sgpr_count", "agpr_count",
                     "group_segment_fixed_size", "private_segment_fixed_size",
                     "vgpr_spill_count", "max_flat_workgroup_size"):
                cur[k] = int(v)
            if k == "symbol":
                cur["symbol"] = v
            if k == "wavefront_size":
                rows.append(cur)
                cur = {}
    for r in rows:
        name = r.get("symbol", r.get("name", "?")).strip("'")
        dem = subprocess.run(["c++filt", name.replace(".kd", "")],
                             stdout=subprocess.PIPE).stdout.decode().strip()
        if pat and not pat.search(dem):
            continue
        print("%4d vgpr %3d agpr %4d sgpr %6d lds %5d scratch %3d spill  %s"
              % (r.get("vgpr_count", -1), r.get("agpr_count", 0),
                 r.get("sgpr_count", -1),
                 r.get("group_segment_fixed_size", 0),
                 r.get("private_segment_fixed_size", 0),
                 r.get("vgpr_spill_count", 0), dem[:150]))


if __name__ == "__main__":
    main()
