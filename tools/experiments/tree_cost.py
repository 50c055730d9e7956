import sys, numpy as np
sys.path.insert(0, '/root/repo')
from androidrenderer_amd import mesh
a = mesh.atrium(int(sys.argv[1]) if len(sys.argv) > 1 else 8).arrays()
pos, idx, prims = a["positions"], a["indices"], a["primitives"]
tris = []
for p in prims:
    i = idx[p["first_index"]:p["first_index"] + p["index_count"]].reshape(-1, 3) + p["vertex_offset"]
    tris.append(pos[i])
T = np.concatenate(tris)  # (n,3,3)
lo, hi = T.min(1), T.max(1)
n = len(T)
print("triangles", n)
def sa(lo, hi):
    d = np.maximum(hi - lo, 0)
    return 2 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0])
def cost(order):
    l, h = lo[order], hi[order]
    root = sa(l.min(0)[None], h.max(0)[None])[0]
    leaf = sa(l, h).sum() / root
    tot = 0.0
    levels = []
    while len(l) > 1:
        m = -(-len(l) // 4)
        pad = m * 4 - len(l)
        if pad:
            l = np.concatenate([l, np.full((pad, 3), np.inf)]); h = np.concatenate([h, np.full((pad, 3), -np.inf)])
        l = l.reshape(m, 4, 3).min(1); h = h.reshape(m, 4, 3).max(1)
        s = sa(l, h).sum() / root
        levels.append(s)
        tot += s
    return tot, leaf, levels
def spread(v):
    v = v.astype(np.uint64)
    out = np.zeros_like(v)
    for b in range(10):
        out |= ((v >> b) & 1) << (3 * b)
    return out
c = (lo + hi) * 0.5
smin, smax = lo.min(0), hi.max(0)
q = np.clip(((c - smin) / (smax - smin) * 1024).astype(np.int64), 0, 1023)
morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
def hilbert(q, bits=10):
    X = q.astype(np.int64).copy().T  # (3,n)
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            sel = (X[i] & Q) != 0
            X[0] = np.where(sel, X[0] ^ P, X[0])
            t = np.where(~sel, (X[0] ^ X[i]) & P, 0)
            X[0] ^= t; X[i] ^= t
        Q >>= 1
    for i in range(1, 3): X[i] ^= X[i - 1]
    t = np.zeros_like(X[0]); Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t); Q >>= 1
    for i in range(3): X[i] ^= t
    key = np.zeros(X.shape[1], np.uint64)
    for b in range(bits):
        for i in range(3):
            key |= ((X[i].astype(np.uint64) >> b) & 1) << (3 * b + (2 - i))
    return key
hk = hilbert(q)
def refine(order, win=64):
    out = order.copy()
    for s in range(0, len(order), win):
        seg = out[s:s + win]
        def rec(seg):
            if len(seg) <= 1: return seg
            cc = c[seg]; ax = np.argmax(cc.max(0) - cc.min(0))
            seg = seg[np.argsort(cc[:, ax], kind="stable")]
            half = len(seg) // 2
            return np.concatenate([rec(seg[:half]), rec(seg[half:])])
        if len(seg) == win: out[s:s + win] = rec(seg)
    return out
def topdown(ids, size):
    # complete 4-ary layout: subtree of `size` (power of 4) slots; split ids into 4 quarters of size/4 slots each (first quarters full)
    if size <= 1 or len(ids) <= 1: return ids
    def split(ids, cap):  # first part gets min(len, cap)
        cc = c[ids]; ax = np.argmax(cc.max(0) - cc.min(0))
        ids = ids[np.argsort(cc[:, ax], kind="stable")]
        k = min(len(ids), cap)
        return ids[:k], ids[k:]
    # balanced variant: distribute evenly instead of filling left first
    half = size // 2; quarter = size // 4
    A, B = split(ids, half)
    parts = []
    for part in (A, B):
        if len(part) == 0: continue
        P, Q = split(part, quarter)
        parts += [topdown(P, quarter), topdown(Q, quarter)] if len(Q) else [topdown(P, quarter)]
    return np.concatenate(parts)
size = 1
while size < n: size *= 4
for name, order in (("input", np.arange(n)), ("morton", np.argsort(morton, kind="stable")), ("hilbert", np.argsort(hk, kind="stable")),
                    ("hilbert+refine64", refine(np.argsort(hk, kind="stable"))), ("hilbert+refine256", refine(np.argsort(hk, kind="stable"), 256)),
                    ("hilbert+refine1024", refine(np.argsort(hk, kind="stable"), 1024)),
                    ("topdown median (left-filled)", topdown(np.arange(n), size))):
    t, leaf, lv = cost(order)
    print(f"{name:32s} internal {t:8.3f} leaf {leaf:7.3f}  levels {[round(x,2) for x in lv]}")

def topdown_sa(ids, size, keyfn=None):
    if size <= 1 or len(ids) <= 1: return ids
    def split(ids, cap):
        if len(ids) <= cap: return ids, ids[:0]
        best = None
        for ax in range(3):
            for kk in ((c[ids][:, ax]), (lo[ids][:, ax]), (hi[ids][:, ax])):
                o = ids[np.argsort(kk, kind="stable")]
                A, B = o[:cap], o[cap:]
                s = sa(lo[A].min(0)[None], hi[A].max(0)[None])[0] * len(A) + sa(lo[B].min(0)[None], hi[B].max(0)[None])[0] * len(B)
                if best is None or s < best[0]: best = (s, A, B)
        return best[1], best[2]
    half = size // 2; quarter = size // 4
    A, B = split(ids, half)
    parts = []
    for part in (A, B):
        if len(part) == 0: continue
        P, Q = split(part, quarter)
        parts += [topdown_sa(P, quarter), topdown_sa(Q, quarter)] if len(Q) else [topdown_sa(P, quarter)]
    return np.concatenate(parts)
o = topdown_sa(np.arange(n), size)
t, leaf, lv = cost(o)
print(f"{'topdown min-SA axis (9 keys)':32s} internal {t:8.3f} leaf {leaf:7.3f}  levels {[round(float(x),2) for x in lv]}")

def refine_sa(order, win):
    out = order.copy()
    for s in range(0, len(order), win):
        seg = out[s:s + win]
        if len(seg) == win: out[s:s + win] = topdown_sa(seg, win)
        else:
            out[s:s + len(seg)] = topdown_sa(seg, win)
    return out
ho = np.argsort(hk, kind="stable")
for win in (64, 256, 1024, 4096, 16384):
    t, leaf, lv = cost(refine_sa(ho, win))
    print(f"{'hilbert + SA refine ' + str(win):32s} internal {t:8.3f}  levels {[round(float(x),2) for x in lv]}")

def topdown_items(ilo, ihi, cnt, ids, size, nkeys, min_size=1):
    # ids: item indices; returns reordered ids.  cost uses SA * (number of triangles)
    if size <= min_size or len(ids) <= 1: return ids
    ic = (ilo + ihi) * 0.5
    def split(ids, cap):
        if len(ids) <= cap: return ids, ids[:0]
        best = None
        for ax in range(3):
            keys = (ic[ids][:, ax],) if nkeys == 3 else (ic[ids][:, ax], ilo[ids][:, ax], ihi[ids][:, ax])
            for kk in keys:
                o = ids[np.argsort(kk, kind="stable")]
                A, B = o[:cap], o[cap:]
                s = sa(ilo[A].min(0)[None], ihi[A].max(0)[None])[0] * cnt[A].sum() + sa(ilo[B].min(0)[None], ihi[B].max(0)[None])[0] * cnt[B].sum()
                if best is None or s < best[0]: best = (s, A, B)
        return best[1], best[2]
    A, B = split(ids, size // 2)
    out = [topdown_items(ilo, ihi, cnt, A, size // 2, nkeys, min_size)]
    if len(B): out.append(topdown_items(ilo, ihi, cnt, B, size // 2, nkeys, min_size))
    return np.concatenate(out)
def multipass(order, clusters, window_items, nkeys, min_size=1):
    order = order.copy()
    for csz in clusters:
        nitems = -(-len(order) // csz)
        ilo = np.full((nitems, 3), np.inf); ihi = np.full((nitems, 3), -np.inf); cnt = np.zeros(nitems)
        for i in range(nitems):
            seg = order[i * csz:(i + 1) * csz]
            ilo[i] = lo[seg].min(0); ihi[i] = hi[seg].max(0); cnt[i] = len(seg)
        perm = np.arange(nitems)
        for s in range(0, nitems, window_items):
            seg = perm[s:s + window_items]
            perm[s:s + len(seg)] = topdown_items(ilo, ihi, cnt, seg, window_items, nkeys, min_size if csz == 1 else 1)
        # note: a partial last cluster must stay last; skip reordering it if partial
        order = np.concatenate([order[i * csz:(i + 1) * csz] for i in perm])
    return order
for nkeys in (3, 9):
    for clusters in ((1,), (16, 1), (64, 1), (256, 16, 1)):
        t, leaf, lv = cost(multipass(ho, clusters, 1024, nkeys, 4))
        print(f"{'multipass ' + str(clusters) + ' keys ' + str(nkeys):36s} internal {t:8.3f}  levels {[round(float(x),2) for x in lv]}")

print("---- input runs as units")
for k in (4, 16, 64, 128, 256, 768):
    nr = -(-n // k)
    rlo = np.array([lo[i*k:(i+1)*k].min(0) for i in range(nr)]); rhi = np.array([hi[i*k:(i+1)*k].max(0) for i in range(nr)])
    rc = (rlo + rhi) * 0.5
    rq = np.clip(((rc - smin) / (smax - smin) * 1024).astype(np.int64), 0, 1023)
    rk = hilbert(rq)
    perm = np.argsort(rk, kind="stable")
    order = np.concatenate([np.arange(i*k, min((i+1)*k, n)) for i in perm])
    t, leaf, lv = cost(order)
    print(f"{'runs of ' + str(k):32s} internal {t:8.3f}  levels {[round(float(x),2) for x in lv]}")
    if k in (64, 256):
        t, leaf, lv = cost(multipass(order, (1,), 1024, 9, 4))
        print(f"{'   + SA refine 1024':32s} internal {t:8.3f}  levels {[round(float(x),2) for x in lv]}")

print("---- fine pass, then coarse passes over the fine pass's clusters")
fine = multipass(ho, (1,), 1024, 9, 4)
for clusters in ((64,), (16,), (256,), (64, 16), (256, 64), (256, 64, 16), (1024, 256, 64)):
    t, leaf, lv = cost(multipass(fine, clusters, 1024, 9, 1))
    print(f"{'fine + coarse ' + str(clusters):36s} internal {t:8.3f}  levels {[round(float(x),2) for x in lv]}")

print("---- fine, coarse, fine again")
for clusters in ((64,), (256, 64), (16,)):
    o2 = multipass(multipass(fine, clusters, 1024, 9, 1), (1,), 1024, 9, 4)
    t, leaf, lv = cost(o2)
    print(f"{'fine + coarse ' + str(clusters) + ' + fine':36s} internal {t:8.3f}  levels {[round(float(x),2) for x in lv]}")
    o3 = multipass(multipass(o2, clusters, 1024, 9, 1), (1,), 1024, 9, 4)
    t, leaf, lv = cost(o3)
    print(f"{'   ... + coarse + fine':36s} internal {t:8.3f}  levels {[round(float(x),2) for x in lv]}")

print("---- Hilbert order of centres displaced along the triangle normal (separates the faces that meet at a seam)")
e1 = T[:, 1] - T[:, 0]; e2 = T[:, 2] - T[:, 0]
nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
ext = float((smax - smin).max())
for frac in (0.0, 0.002, 0.005, 0.01, 0.02, 0.05):
    cd = c + nrm * (frac * ext)
    qd = np.clip(((cd - smin) / (smax - smin) * 1024).astype(np.int64), 0, 1023)
    od = np.argsort(hilbert(qd), kind="stable")
    t0, _, _ = cost(od)
    t1, _, lv = cost(multipass(od, (1,), 1024, 9, 4))
    print(f"displacement {frac:6.3f} x extent: curve {t0:7.3f}   + SA refine 1024 {t1:7.3f}  levels {[round(float(x),2) for x in lv]}")
