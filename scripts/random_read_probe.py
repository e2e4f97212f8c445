import sys, ctypes as C
sys.path.insert(0,'.')
import vstree_amd as V
for size in (float(a) for a in (sys.argv[1:] or ['32e9', '3e9'])):
    for inf in (1,4,8):
        g=C.c_double()
        V._check(V.lib.vsa_measure_random_read(int(size), inf, 0, C.byref(g)))
        print("table %.0f GB, %d in flight: %.1f G reads/s = %.2f TB/s of 64-byte sectors" % (size/1e9, inf, g.value, g.value*64/1e3), flush=True)
