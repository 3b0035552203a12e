"""Which order of float64 operations NumPy's small products take on this host (the device mirrors them):
(3,3) @ (3,) [dgemv], (3,3) @ (3,3) and (3,3) @ (3,n) [dgemm], 3-vector add / subtract.  Prints how many of
2000 random cases each candidate reproduces bit for bit."""
import math
import numpy as np

rng = np.random.default_rng(0)
fma = math.fma if hasattr(math, "fma") else None


def fma_(a, b, c):
    if fma:
        return fma(a, b, c)
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.fma.restype = ctypes.c_double
    libm.fma.argtypes = [ctypes.c_double] * 3
    return libm.fma(a, b, c)


def cands(r, v):
    a, b, c = float(r[0]), float(r[1]), float(r[2])
    x, y, z = float(v[0]), float(v[1]), float(v[2])
    return {
        "plain ((a x + b y) + c z)": (a * x + b * y) + c * z,
        "fma chain from left fma(c,z,fma(b,y,a x))": fma_(c, z, fma_(b, y, a * x)),
        "fma chain from right fma(a,x,fma(b,y,c z))": fma_(a, x, fma_(b, y, c * z)),
        "plain (a x + (b y + c z))": a * x + (b * y + c * z),
        "fma(a,x, b y) + c z": fma_(a, x, b * y) + c * z,
        "fma(b,y,a x) + c z": fma_(b, y, a * x) + c * z,
    }


def probe(kind):
    hits = {}
    n = 2000
    for _ in range(n):
        R = rng.normal(size=(3, 3))
        if kind == "matvec":
            v = rng.normal(size=3) * 10
            got = R @ v
            rows = [(R[i], v, got[i]) for i in range(3)]
        elif kind == "matvec_T":
            v = rng.normal(size=3) * 10
            got = (R @ v.T).T
            rows = [(R[i], v, got[i]) for i in range(3)]
        elif kind == "mat33":
            Q = rng.normal(size=(3, 3))
            got = R @ Q
            rows = [(R[i], Q[:, j], got[i, j]) for i in range(3) for j in range(3)]
        else:
            P = rng.normal(size=(1000, 3)) * 30
            got = R @ P.T
            rows = [(R[i], P[j], got[i, j]) for i in range(3) for j in (0, 17, 999)]
        for r, v, g in rows:
            for name, val in cands(r, v).items():
                hits.setdefault(name, 0)
                hits[name] += int(val == g)
        total = len(rows) * n
    print(kind)
    for name, h in hits.items():
        print("   %-46s %7d / %d" % (name, h, len(rows) * n))


for k in ("matvec", "matvec_T", "mat33", "mat3n"):
    probe(k)


def probe_matvec_all():
    """Every way of combining the three products of a row of (3,3) @ (3,) with at most two fused steps."""
    import itertools
    hits, n = {}, 3000
    for _ in range(n):
        R, v = rng.normal(size=(3, 3)), rng.normal(size=3) * 10
        got = R @ v
        for row in range(3):
            t = [(float(R[row][k]), float(v[k])) for k in range(3)]
            for i, j, k in itertools.permutations(range(3)):
                (a, x), (b, y), (c, z) = t[i], t[j], t[k]
                for name, val in ((f"(p{i}+p{j})+p{k}", (a * x + b * y) + c * z), (f"fma{k}(p{i}+p{j})", fma_(c, z, a * x + b * y)),
                                  (f"fma{k}(fma{j}(p{i}))", fma_(c, z, fma_(b, y, a * x))), (f"fma{j}(p{i})+p{k}", fma_(b, y, a * x) + c * z)):
                    hits[name] = hits.get(name, 0) + int(val == got[row])
    print("matvec, all orders (best 4 of %d rows)" % (3 * n))
    for name, h in sorted(hits.items(), key=lambda kv: -kv[1])[:4]:
        print("   %-24s %6d" % (name, h))


probe_matvec_all()
import numpy
print(numpy.__version__)
try:
    numpy.show_config()
except Exception as e:
    print(e)
