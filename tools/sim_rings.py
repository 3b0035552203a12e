"""Ring visits of the search on the real KITTI pair 0 -> 1, simulated on the CPU from the restatement's per-ring results:
(1) the all-rings pass with its outliers moved to a queue after +-K lock-step passes, (2) the hinted pass as built, with perfect
knowledge and without pruning, (3) the hinted pass with sharper per-bin bounds (finer azimuth bins, range bands).
    python tools/sim_rings.py [every_nth_wave]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_oracle as O
from pointcloudprocessing_amd import lidar
NB = 64
step = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'kitti_pair.npz'))
a, b = lidar.frame_from_raw(g["raw0"]), lidar.frame_from_raw(g["raw1"])
o = np.zeros(3)
lines = O.get_scan_lines(a, o)
Z = len(lines)
a4, b4, o4 = O.pad_points(a), O.pad_points(b), O.origin4(o)
ua = O.unit_sphere(a4, o4).astype(np.float64)
elev_a = np.arcsin(np.clip(ua[:, 1], -1, 1))
bin_a = np.clip(((np.arctan2(ua[:, 2], ua[:, 0]) + np.pi) * NB / (2 * np.pi)).astype(int), 0, NB - 1)
ring_lo, ring_hi = np.zeros(Z), np.zeros(Z)
be_lo = np.full((Z, NB), np.inf); be_hi = np.full((Z, NB), -np.inf)
for r, (s, e) in enumerate(lines):
    sl = slice(s, e + 1)
    ring_lo[r], ring_hi[r] = elev_a[sl].min(), elev_a[sl].max()
    lo1 = np.full(NB, np.inf); hi1 = np.full(NB, -np.inf)
    np.minimum.at(lo1, bin_a[sl], elev_a[sl]); np.maximum.at(hi1, bin_a[sl], elev_a[sl])
    be_lo[r] = np.minimum(lo1, np.minimum(np.roll(lo1, 1), np.roll(lo1, -1)))
    be_hi[r] = np.maximum(hi1, np.maximum(np.roll(hi1, 1), np.roll(hi1, -1)))
n_waves = len(b) // 64
waves = np.arange(0, n_waves, step)
qi = (waves[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
q4 = b4[qi]
ref = O.unit_sphere(q4, o4)
vq = q4.astype(np.float64) - np.array([o4[0], o4[1] + 0.127, o4[2], o4[3]])
rho_q = np.linalg.norm(vq, axis=1)
elev_q = np.arcsin(np.clip(ref[:, 1].astype(np.float64), -1, 1))
bin_q = np.clip(((np.arctan2(ref[:, 2].astype(np.float64), ref[:, 0].astype(np.float64)) + np.pi) * NB / (2 * np.pi)).astype(int), 0, NB - 1)
rxz = np.hypot(vq[:, 0], vq[:, 2])
far = (rxz * np.sin(0.99 * 2 * np.pi / NB)) ** 2
scan = np.asarray(lines, dtype=np.int64)
D = np.zeros((len(qi), Z))
U = O.unit_sphere(a4, o4)
for r in range(Z):
    beg = np.full(len(qi), scan[r, 0]); end = np.full(len(qi), scan[r, 1])
    closest = O.ring_search(U, ref, beg, end)
    _, d = O.window_min(a4, q4, beg, end, closest, 10)
    D[:, r] = d
def gap(x, lo, hi):
    return np.maximum(0.0, np.maximum(lo - x, x - hi))
whole = (rho_q[:, None] * np.sin(np.minimum(gap(elev_q[:, None], ring_lo[None, :], ring_hi[None, :]), np.pi / 2))) ** 2
de = np.minimum(gap(elev_q[:, None], be_lo[:, bin_q].T, be_hi[:, bin_q].T), np.pi / 2)
none = ~np.isfinite(be_lo[:, bin_q].T)
near_elev = np.where(none, np.inf, (rho_q[:, None] * np.sin(de)) ** 2)
lb = np.maximum(whole, np.minimum(near_elev, far[:, None]))
centre = np.argmin(whole, axis=1)
Q = len(qi)
W = len(waves)
def walk(KA):
    mind = np.full(Q, np.inf)
    passes_A = np.zeros(W, dtype=int); visits = np.zeros(Q, dtype=int)
    # phase A: k in [0, KA]
    wants_later = np.zeros(Q, dtype=bool)
    rec = []
    for k in range(2 * Z + 1):
        off = (k + 1) // 2 if (k & 1) else -(k // 2)
        r = centre + off
        ok = (r >= 0) & (r < Z)
        rr = np.clip(r, 0, Z - 1)
        want = ok & ((mind == np.inf) | (lb[np.arange(Q), rr] <= mind))
        d = D[np.arange(Q), rr]
        mind = np.where(want & (d < mind), d, mind)
        rec.append(want)
        visits += want
    rec = np.array(rec)            # (K, Q)
    passA = rec[:KA + 1].reshape(KA + 1, W, 64).any(axis=2).sum(axis=0)     # per wave
    later = rec[KA + 1:]
    outlier = later.any(axis=0)
    # phase B: compact outliers in order, 64 per wave; passes = for each compacted wave, number of k with any want
    idx = np.nonzero(outlier)[0]
    nB = -(-len(idx) // 64)
    passB = 0
    for w in range(nB):
        sel = idx[w * 64:(w + 1) * 64]
        passB += later[:, sel].any(axis=1).sum()
    total_now = rec.reshape(rec.shape[0], W, 64).any(axis=2).sum()
    print(f"KA={KA:3d}: now {total_now / W:6.2f} passes/wave; A {passA.sum() / W:5.2f} + B {passB / W:5.2f} = {(passA.sum() + passB) / W:5.2f} "
          f"outliers {outlier.mean() * 100:5.2f}% of queries, in {outlier.reshape(W, 64).any(axis=1).mean() * 100:5.1f}% of waves; visits/query {visits.mean():.2f}"
          f"; B waves {nB} lanes/pass B {later[:, idx].sum() / max(passB,1):.1f}")
for KA in (2, 4, 6, 8, 12, 16):
    walk(KA)
print("---- hinted iteration (same clouds, hint = winning ring of the all-rings pass)")
h = np.argmin(D, axis=1)     # ties: lowest ring, fine for stats
def hinted(lbm, name):
    mind = np.full(Q, np.inf)
    rec = []
    for k in range(5):
        off = (k + 1) // 2 if (k & 1) else -(k // 2)
        r = h + off
        ok = (r >= np.maximum(0, h - 2)) & (r < np.minimum(Z, h + 2))
        rr = np.clip(r, 0, Z - 1)
        want = ok & ((mind == np.inf) | (lbm[np.arange(Q), rr] <= mind))
        d = D[np.arange(Q), rr]
        mind = np.where(want & (d < mind), d, mind)
        rec.append(want)
    rec = np.array(rec)
    passes = rec.reshape(5, W, 64).any(axis=2).sum()
    print(f"{name:30s}: visits/query {rec.sum() / Q:.2f}, passes/wave {passes / W:.2f}, lanes/pass {rec.sum() / passes:.1f}; per k: {[round(float(x),2) for x in rec.mean(axis=1)]}")
hinted(lb, "as built")
hinted(D, "perfect knowledge")
hinted(np.zeros_like(D), "no pruning")
print("dist of winning d (m): median %.3f, 90th %.3f, 99th %.3f" % tuple(np.sqrt(np.percentile(D.min(axis=1), [50, 90, 99]))))
print("---- sharper bounds for the hinted iteration")
va = a4.astype(np.float64) - np.array([o4[0], o4[1] + 0.127, o4[2], o4[3]])
rho_a = np.linalg.norm(va, axis=1)
az_a = np.arctan2(ua[:, 2], ua[:, 0]) + np.pi
az_q = np.arctan2(ref[:, 2].astype(np.float64), ref[:, 0].astype(np.float64)) + np.pi
def bands(nb, width, use_range):
    ba = np.clip((az_a * nb / (2 * np.pi)).astype(int), 0, nb - 1)
    bq = np.clip((az_q * nb / (2 * np.pi)).astype(int), 0, nb - 1)
    e_lo = np.full((Z, nb), np.inf); e_hi = np.full((Z, nb), -np.inf)
    r_lo = np.full((Z, nb), np.inf); r_hi = np.full((Z, nb), -np.inf)
    for r, (s, e) in enumerate(lines):
        sl = slice(s, e + 1)
        for arr_lo, arr_hi, val in ((e_lo, e_hi, elev_a[sl]), (r_lo, r_hi, rho_a[sl])):
            lo1 = np.full(nb, np.inf); hi1 = np.full(nb, -np.inf)
            np.minimum.at(lo1, ba[sl], val); np.maximum.at(hi1, ba[sl], val)
            arr_lo[r] = np.min([np.roll(lo1, k) for k in range(-width, width + 1)], axis=0)
            arr_hi[r] = np.max([np.roll(hi1, k) for k in range(-width, width + 1)], axis=0)
    d_e = np.minimum(gap(elev_q[:, None], e_lo[:, bq].T, e_hi[:, bq].T), np.pi / 2)
    nn = ~np.isfinite(e_lo[:, bq].T)
    s2 = np.sin(d_e) ** 2
    with np.errstate(invalid="ignore"):
        if use_range:
            rs = np.clip(rho_q[:, None] * (1 - s2 / 2), r_lo[:, bq].T, r_hi[:, bq].T)
            near = np.where(nn, np.inf, (rho_q[:, None] - rs) ** 2 + rho_q[:, None] * rs * s2)
        else:
            near = np.where(nn, np.inf, (rho_q[:, None] * np.sin(d_e)) ** 2)
    ang = min(0.99 * width * 2 * np.pi / nb, np.pi / 2)
    return np.minimum(near, ((rxz * np.sin(ang)) ** 2)[:, None])
for nb, width, rng in ((64, 1, False), (64, 1, True), (256, 1, False), (256, 1, True), (1024, 1, True), (1024, 2, True), (2048, 2, True)):
    hinted(np.maximum(whole, bands(nb, width, rng)), f"bins {nb} width {width} range {rng}")
print("---- the block-shared search: hinted ring first, the other rings of the range listed against that first distance only")
def listed(lbm, name, block=256):
    d0 = D[np.arange(Q), h]
    items = np.zeros(Q)
    for off in (1, -1, -2):
        r = h + off
        ok = (r >= np.maximum(0, h - 2)) & (r < np.minimum(Z, h + 2))
        rr = np.clip(r, 0, Z - 1)
        items += ok & (lbm[np.arange(Q), rr] <= d0)
    nblk = Q // block
    per_block = items[:nblk * block].reshape(nblk, block).sum(axis=1)
    waves_per_block = block // 64
    passes = waves_per_block + np.ceil(per_block / 64)
    print(f"{name:34s}: listed rings/query {items.mean():.2f}, wave passes per {block} queries {passes.mean():.2f} (lock step as built: see above x {waves_per_block})")
def simple_range(nb):
    """per (ring, bin +-1) the range band; bound = (distance of the query's range from the band)^2"""
    ba = np.clip((az_a * nb / (2 * np.pi)).astype(int), 0, nb - 1)
    bq = np.clip((az_q * nb / (2 * np.pi)).astype(int), 0, nb - 1)
    r_lo = np.full((Z, nb), np.inf); r_hi = np.full((Z, nb), -np.inf)
    for r, (s, e) in enumerate(lines):
        sl = slice(s, e + 1)
        lo1 = np.full(nb, np.inf); hi1 = np.full(nb, -np.inf)
        np.minimum.at(lo1, ba[sl], rho_a[sl]); np.maximum.at(hi1, ba[sl], rho_a[sl])
        r_lo[r] = np.min([np.roll(lo1, k) for k in (-1, 0, 1)], axis=0)
        r_hi[r] = np.max([np.roll(hi1, k) for k in (-1, 0, 1)], axis=0)
    g = gap(rho_q[:, None], r_lo[:, bq].T, r_hi[:, bq].T)
    return np.where(np.isfinite(g), g, 0.0) ** 2
listed(lb, "as built")
rb = simple_range(64)
near_both = np.maximum(near_elev, np.where(none, np.inf, rb))
listed(np.maximum(whole, np.minimum(near_both, far[:, None])), "plus range band (simple gap)")
listed(np.maximum(whole, bands(64, 1, True)), "plus range band (joint bound)")
listed(D, "perfect knowledge")
hinted(np.maximum(whole, np.minimum(near_both, far[:, None])), "lock step, simple range band")
print("---- the all-rings walk with the range bands in its bounds")
lb_saved = lb
lb = np.maximum(whole, np.minimum(near_both, far[:, None]))
walk(8)
lb = lb_saved
print("---- the all-rings search as a block-shared list: start ring first, every other ring listed against that first distance")
def listed_all(lbm, name, block=256):
    d0 = D[np.arange(Q), centre]
    cand = lbm <= d0[:, None]
    cand[np.arange(Q), centre] = False
    items = cand.sum(axis=1)
    nblk = Q // block
    per_block = items[:nblk * block].reshape(nblk, block).sum(axis=1)
    passes = block // 64 + np.ceil(per_block / 64)
    print(f"{name:34s}: listed rings/query {items.mean():.2f}, wave passes per wave {passes.mean() / (block // 64):.2f}")
listed_all(lb_saved, "as built (elevation bands)")
listed_all(np.maximum(whole, np.minimum(near_both, far[:, None])), "plus range bands")
d0 = D[np.arange(Q), centre]
def bound_gap(g, rho):
    return (rho * np.sin(np.minimum(np.maximum(g, 0.0), np.pi / 2))) ** 2
print("---- the all-rings walk confined by cumulative bands (ring 0 is the highest ring): rings left between the two cut-offs")
pmin_lo = np.minimum.accumulate(ring_lo)                        # lowest elevation among rings 0 .. r
smax_hi = np.maximum.accumulate(ring_hi[::-1])[::-1]            # highest elevation among rings r .. Z-1
for name, use_d in (("after the first pass (its distance)", d0), ("with the final distance", D.min(axis=1))):
    width = np.zeros(Q, dtype=int)
    for qq in range(Q):
        below = bound_gap(pmin_lo - elev_q[qq], rho_q[qq]) > use_d[qq]     # all rings <= r ruled out
        above = bound_gap(elev_q[qq] - smax_hi, rho_q[qq]) > use_d[qq]     # all rings >= r ruled out
        lo_cut = np.nonzero(below)[0].max() + 1 if below.any() else 0
        hi_cut = np.nonzero(above)[0].min() if above.any() else Z
        width[qq] = max(1, hi_cut - lo_cut)
    per_wave = width.reshape(W, 64).max(axis=1)
    print("%s: rings left per query mean %.1f, median %d, 90th %d, 99th %d; per wave (widest lane) mean %.1f" %
          (name, width.mean(), np.median(width), np.percentile(width, 90), np.percentile(width, 99), per_wave.mean()))
print("---- listed rings with each of the three bins bounded on its own (the neighbours also by the azimuth gap to their edge)")
def own_bands(nb):
    ba = np.clip((az_a * nb / (2 * np.pi)).astype(int), 0, nb - 1)
    e_lo = np.full((Z, nb), np.inf); e_hi = np.full((Z, nb), -np.inf); r_lo = np.full((Z, nb), np.inf); r_hi = np.full((Z, nb), -np.inf)
    for r, (s_, e_) in enumerate(lines):
        sl = slice(s_, e_ + 1)
        np.minimum.at(e_lo[r], ba[sl], elev_a[sl]); np.maximum.at(e_hi[r], ba[sl], elev_a[sl])
        np.minimum.at(r_lo[r], ba[sl], rho_a[sl]); np.maximum.at(r_hi[r], ba[sl], rho_a[sl])
    return e_lo, e_hi, r_lo, r_hi
def three_bins(nb, union):
    e_lo, e_hi, r_lo, r_hi = own_bands(nb)
    w = 2 * np.pi / nb
    bq = np.clip((az_q * nb / (2 * np.pi)).astype(int), 0, nb - 1)
    off = az_q - bq * w                                              # position inside the bin, 0 .. w
    out = np.full((Q, Z), np.inf)
    def one(bins_idx, lateral):
        el = e_lo[:, bins_idx].T; eh = e_hi[:, bins_idx].T; rl = r_lo[:, bins_idx].T; rh = r_hi[:, bins_idx].T
        none_ = ~np.isfinite(el)
        de = np.minimum(gap(elev_q[:, None], el, eh), np.pi / 2)
        be = (rho_q[:, None] * np.sin(de)) ** 2
        g = gap(rho_q[:, None], rl, rh)
        br = np.where(np.isfinite(g), g, 0.0) ** 2
        b_ = np.maximum(np.maximum(be, br), lateral[:, None])
        return np.where(none_, np.inf, b_)
    zero = np.zeros(Q)
    if union:                                                        # as built: one band over the three bins
        return None
    lat_lo = (rxz * np.sin(np.minimum(0.99 * off, np.pi / 2))) ** 2          # points of bin b-1 lie beyond the lower edge
    lat_hi = (rxz * np.sin(np.minimum(0.99 * (w - off), np.pi / 2))) ** 2
    b0 = one(bq, zero); bm = one((bq - 1) % nb, lat_lo); bp = one((bq + 1) % nb, lat_hi)
    far_ = (rxz * np.sin(min(0.99 * w, np.pi / 2))) ** 2
    return np.maximum(whole, np.minimum(np.minimum(b0, np.minimum(bm, bp)), far_[:, None]))
def union_bands(nb):
    e_lo, e_hi, r_lo, r_hi = own_bands(nb)
    bq = np.clip((az_q * nb / (2 * np.pi)).astype(int), 0, nb - 1)
    roll3 = lambda x, f: f(f(x, np.roll(x, 1, axis=1)), np.roll(x, -1, axis=1))
    el = roll3(e_lo, np.minimum)[:, bq].T; eh = roll3(e_hi, np.maximum)[:, bq].T
    rl = roll3(r_lo, np.minimum)[:, bq].T; rh = roll3(r_hi, np.maximum)[:, bq].T
    none_ = ~np.isfinite(el)
    be = (rho_q[:, None] * np.sin(np.minimum(gap(elev_q[:, None], el, eh), np.pi / 2))) ** 2
    g = gap(rho_q[:, None], rl, rh); br = np.where(np.isfinite(g), g, 0.0) ** 2
    far_ = (rxz * np.sin(min(0.99 * 2 * np.pi / nb, np.pi / 2))) ** 2
    return np.maximum(whole, np.minimum(np.where(none_, np.inf, np.maximum(be, br)), far_[:, None]))
for nb in (128, 256):
    listed(union_bands(nb), f"{nb} bins, one band over three bins")
    listed(three_bins(nb, False), f"{nb} bins, three bins on their own")
