"""Random-read rate over the tables of a live 3 Gbp index, where the driver put
them (vsa_measure_table_read), next to the same probe on a fresh allocation of
the same size (vsa_measure_random_read)."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import vstree_amd as V

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
dg = V.device_malloc(n + 64, 0)
V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, n, dg, 0))
t0 = time.time()
index = V.Index.build_device(dg, n, 4, 0, 0)
V.device_free(dg, 0)
info = index.info()
print("index %d bp built in %.1f s, %.1f GB" % (n, time.time() - t0,
                                                info.device_bytes / 1e9))
g = C.c_double()
for table, name in ((0, "slot16"), (1, "esa8"), (2, "tis2"), (3, "suf"),
                    (4, "slot16/8B"), (5, "esa8/16B")):
    for inflight in (1, 4):
        rc = V.lib.vsa_measure_table_read(index._h, table, inflight,
                                          C.byref(g))
        print("%-11s inflight %d  rc %d  %.1f G reads/s" % (name, inflight, rc,
                                                           g.value))
index.close()
for size in (24e9, 69e9):
    for inflight in (1, 4):
        V._check(V.lib.vsa_measure_random_read(int(size), inflight, 0,
                                               C.byref(g)))
        print("fresh %3.0f GB inflight %d  %.1f G reads/s" % (size / 1e9,
                                                             inflight, g.value))
