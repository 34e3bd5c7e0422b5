"""Does a fixed relative offset between a, b and c inside ONE slab move the f32-add rate?

a = slab, b = slab + 1 GiB + db, c = slab + 2 GiB + dc, for db, dc over a ladder of offsets (4 KiB .. 64 MiB).
If some (db, dc) is reproducibly faster across processes, the pool can colour large blocks with it.
"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import simplemath_amd as sma
lib = sma.load()
n = 1 << 28
GiB = 1 << 30
KiB = 1 << 10
MiB = 1 << 20
fn = lib.c.smhip_contiguous
def rate(a, b, c, steps=20):
    args = (C.c_int(0), C.c_int(0), C.c_void_p(a), C.c_void_p(b), C.c_void_p(c), C.c_size_t(n))
    for _ in range(2): fn(*args)
    e0, e1 = lib.event(), lib.event()
    lib.synchronize(); lib.record(e0)
    for _ in range(steps): fn(*args)
    lib.record(e1); lib.synchronize()
    return 12.0 * n / (lib.elapsed_ms(e0, e1) / steps) * 1e-6
slab = lib.alloc(3 * GiB + 256 * MiB)
lib.c.smhip_fill_uniform_f32(C.c_void_p(slab), C.c_size_t((3 * GiB + 256 * MiB) // 4), 1, 0, C.c_float(-1.0), C.c_float(1.0))
for _ in range(200): rate(slab, slab + GiB, slab + 2 * GiB, steps=2)   # ramp clocks
offs = [0, 4 * KiB, 16 * KiB, 64 * KiB, 256 * KiB, 1 * MiB, 2 * MiB, 3 * MiB, 4 * MiB, 6 * MiB, 8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB]
def label(o): return ("%dK" % (o // KiB)) if o < MiB else ("%dM" % (o // MiB))
print("rows: db (offset of b past slab+1GiB); cols: dc (offset of c past slab+2GiB); GB/s")
print("%6s " % "" + " ".join("%5s" % label(o) for o in offs))
best = (0, 0, 0)
for db in offs:
    row = []
    for dc in offs:
        r = rate(slab, slab + GiB + db, slab + 2 * GiB + dc)
        row.append(r)
        if r > best[0]: best = (r, db, dc)
    print("%6s " % label(db) + " ".join("%5.0f" % r for r in row), flush=True)
print("best %.0f GB/s at db=%s dc=%s" % (best[0], label(best[1]), label(best[2])))
# re-measure: baseline, best, baseline, best
for k in range(3):
    print("recheck %d: base %.0f  best %.0f" % (k, rate(slab, slab + GiB, slab + 2 * GiB, 40), rate(slab, slab + GiB + best[1], slab + 2 * GiB + best[2], 40)), flush=True)
