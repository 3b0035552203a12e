"""All-rings first iteration on the real pair: lock-step passes over the first offsets, then every lane's remaining rings as a
static list pooled into dense passes (CPU simulation on top of sim_rings.py's set-up):  python tools/sim_pool.py"""
import os, sys
import numpy as np
sys.argv = ['sim', '4']
# run the simulator's set-up only: execute up to the definition of walk()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sim_rings.py')).read()
head = src[:src.index("def walk(KA):")]
ns = {'__file__': os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sim_rings.py')}
exec(compile(head, 'sim_head', 'exec'), ns)
Q, W, Z, D, lb, centre = ns['Q'], ns['W'], ns['Z'], ns['D'], ns['lb'], ns['centre']
ar = np.arange(Q)
def pooled(KA, waves_per_block):
    mind = np.full(Q, np.inf)
    passA = 0; visA = 0
    for k in range(KA + 1):
        off = (k + 1) // 2 if (k & 1) else -(k // 2)
        r = centre + off
        ok = (r >= 0) & (r < Z)
        rr = np.clip(r, 0, Z - 1)
        want = ok & ((mind == np.inf) | (lb[ar, rr] <= mind))
        d = D[ar, rr]
        mind = np.where(want & (d < mind), d, mind)
        passA += want.reshape(W, 64).any(axis=1).sum(); visA += want.sum()
    # static list: every ring beyond the phase-A offsets whose bound does not exceed mind
    offs = np.arange(Z)[None, :] - centre[:, None]
    kk = np.where(offs > 0, 2 * offs - 1, -2 * offs)
    todo = (kk > KA) & (lb <= mind[:, None])
    items = todo.sum(axis=1)
    nblk = W // waves_per_block
    per_block = items[:nblk * waves_per_block * 64].reshape(nblk, -1).sum(axis=1)
    passB = np.ceil(per_block / 64).sum() / (nblk * waves_per_block)
    print(f"KA={KA}: A {passA / W:.2f} passes/wave ({visA / Q:.2f} visits/query) + static list {items.mean():.2f} visits/query "
          f"-> pooled over {waves_per_block} wave(s): {passB:.2f} passes/wave; total {passA / W + passB:.2f} (now 10.0); "
          f"longest list {items.max()}, 99th pct {np.percentile(items, 99):.0f}")
for KA in (0, 2, 4):
    for wpb in (1, 4):
        pooled(KA, wpb)
