"""CPU model of the pair-lane kNN kernel (glim_amd/csrc/knn_pairs.hip): 32 queries per wavefront, two lanes per query with separate top-k lists over
disjoint candidates and a shared pruning bound, lists merged at the end.  Like tools/knn_model.py it follows the kernel step by step and counts the
lock-step insertion rounds; it exists to check, without a GPU, that the per-lane threshold selection of the chunk kernels leaves the merged
lists exact in THIS kernel too (a lane selects over its own list only) and to price it.  A design tool, not an oracle.

  python tools/knn_pair_model.py        # 65 536-pt scan: rounds with / without the selection, lists compared with the exhaustive answer
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from knn_model import hilbert_keys  # noqa: E402

Q, K = 32, 10
f32 = np.float32


class Lists:
    def __init__(self, self_idx):
        self.d = np.full((Q, K), np.inf)
        self.i = np.repeat(self_idx[:, None], K, axis=1).astype(np.int64)

    def push(self, lanes, dn, idn):
        if len(lanes) == 0:
            return
        d, i = self.d[lanes], self.i[lanes]
        ok = (dn < d[:, -1]) | ((dn == d[:, -1]) & (idn < i[:, -1]))
        if not ok.any():
            return
        lanes, d, i, dn, idn = lanes[ok], d[ok], i[ok], dn[ok], idn[ok]
        pos = ((d < dn[:, None]) | ((d == dn[:, None]) & (i < idn[:, None]))).sum(axis=1)
        cols = np.arange(K)[None, :]
        self.d[lanes] = np.where(cols < pos[:, None], d, np.where(cols == pos[:, None], dn[:, None], np.roll(d, 1, axis=1)))
        self.i[lanes] = np.where(cols < pos[:, None], i, np.where(cols == pos[:, None], idn[:, None], np.roll(i, 1, axis=1)))


def run(pts, select=None, bits=None):
    """select: None or (bisection steps, minimum accepted count that triggers a selection)."""
    n = len(pts)
    keys = hilbert_keys(pts, bits if bits else (8 if n < 32768 else 13))
    perm = np.argsort(keys, kind="stable")
    C64 = (n + 63) // 64
    C = 2 * C64
    spt = np.full((C * Q, 3), np.inf)
    spt[:n] = pts[perm].astype(np.float64)
    sidx = np.full(C * Q, -1, dtype=np.int64)
    sidx[:n] = perm
    P, I = spt.reshape(C, Q, 3), sidx.reshape(C, Q)
    live = I >= 0
    lo = np.where(live[..., None], P, np.inf).min(axis=1).astype(f32)
    hi = np.where(live[..., None], P, -np.inf).max(axis=1).astype(f32)
    G = (C + 63) // 64
    lanes = np.arange(Q)
    rounds = np.zeros(C, dtype=np.int64)
    out = np.zeros((n, K), dtype=np.int64)
    nseed = min(K + 2, Q)
    for c in range(C):
        if not live[c].any():
            continue
        q = np.where(live[c][:, None], P[c], P[c][0])
        L = [Lists(np.where(live[c], I[c], -1)), Lists(np.where(live[c], I[c], -1))]

        def scan_pair(ccs, needs, seed):
            seeded = np.zeros((Q, Q), dtype=bool)
            if seed:
                for t in range(nseed):
                    off = ((t + 1) >> 1) if (t & 1) else -(t >> 1)
                    j = (lanes + off) & (Q - 1)
                    seeded[lanes, j] = True
                    ok = lanes[I[ccs[0]][j] >= 0]
                    dj = (q[ok] - P[ccs[0]][j[ok]]) ** 2
                    L[0].push(ok, (dj[:, 0] + dj[:, 1]) + dj[:, 2], I[ccs[0]][j[ok]])
            shared = np.minimum(L[0].d[:, -1], L[1].d[:, -1])
            work = []
            for h in (0, 1):
                cc = ccs[h]
                if cc < 0:
                    work.append(None)
                    continue
                thr = np.where(needs[h], shared, -1.0)
                dd = q[:, None, :] - P[cc][None, :, :]
                dall = (dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1]) + dd[..., 2] * dd[..., 2]
                d32 = dall.astype(f32)
                thr32 = ((thr * 1.000002).astype(f32) + f32(1e-37)).astype(f32)
                m = d32 <= thr32[:, None]
                if h == 0:
                    m &= ~seeded
                if select is not None and m.sum(axis=1).max() > select[1]:
                    dv = np.where(m, d32, f32(np.inf))

                    def count_le(t):
                        return (dv <= t[:, None]).sum(axis=1) + (L[h].d <= t.astype(np.float64)[:, None]).sum(axis=1)

                    hi_b = np.minimum(thr32, f32(3.4028234e38)).view(np.uint32).astype(np.int64)
                    hi_b = np.where(thr32 < 0, 0, hi_b)
                    sel = needs[h] & (count_le(hi_b.astype(np.uint32).view(f32)) >= K)
                    lo_b = np.where(hi_b > (16 << 23), hi_b - (16 << 23), 0)
                    for _ in range(select[0]):
                        mid = lo_b + ((hi_b - lo_b) >> 1)
                        ok = count_le(mid.astype(np.uint32).view(f32)) >= K
                        hi_b = np.where(ok, mid, hi_b)
                        lo_b = np.where(ok, lo_b, mid + 1)
                    with np.errstate(over="ignore"):
                        keep = (hi_b.astype(np.uint32).view(f32) * f32(1.000002) + f32(1e-37)).astype(f32)
                    m[sel] &= (dv <= keep[:, None])[sel]
                work.append((cc, m, dall))
            cnts = [w[1].sum(axis=1) if w else np.zeros(Q, dtype=np.int64) for w in work]
            r = int(max(cnts[0].max(), cnts[1].max()))
            rounds[c] += r
            for h in (0, 1):
                if not work[h]:
                    continue
                cc, m, dall = work[h]
                jj = np.argsort(~m, axis=1, kind="stable")
                for t in range(int(cnts[h].max())):
                    ln = lanes[cnts[h] > t]
                    j = jj[ln, t]
                    good = I[cc][j] >= 0
                    L[h].push(ln[good], dall[ln[good], j[good]], I[cc][j[good]])

        ones = np.ones(Q, bool)
        scan_pair((c, c + 1 if c + 1 < C else -1), (ones, ones), True)
        scan_pair((c - 1 if c > 0 else -1, c + 2 if c + 2 < C else -1), (ones, ones), False)
        gc = c // 64
        for t in range(2 * G):
            gi = gc + ((t + 1) >> 1) if (t & 1) else gc - (t >> 1)
            if gi < 0 or gi >= G:
                continue
            pm = np.minimum(L[0].d[:, -1], L[1].d[:, -1])
            R = f32(np.sqrt(pm.max()) * 1.000001) + f32(1e-30)
            cands = np.arange(gi * 64, min((gi + 1) * 64, C))
            cands = cands[(cands < c - 1) | (cands > c + 2)]
            ok = np.ones(len(cands), bool)
            for a in range(3):
                ok &= (lo[cands, a] <= hi[c, a] + R) & (hi[cands, a] >= lo[c, a] - R)
            cands = list(cands[ok])
            while cands:
                cc_a = cands.pop(0)
                cc_b = cands.pop(0) if cands else -1
                needs = []
                for h, cc in ((0, cc_a), (1, cc_b)):
                    if cc < 0:
                        needs.append(np.zeros(Q, bool))
                        continue
                    g = np.maximum(0.0, np.maximum(lo[cc].astype(np.float64) - q, q - hi[cc].astype(np.float64)))
                    cur = np.minimum(L[h].d[:, -1], pm)
                    needs.append(((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]) * (1.0 - 1e-12) <= cur)
                if not (needs[0].any() or needs[1].any()):
                    continue
                scan_pair((cc_a, cc_b), needs, False)
        # merge: the k best of the two lists of every query (placeholders are (+inf, self) in both)
        d = np.concatenate([L[0].d, L[1].d], axis=1)
        i = np.concatenate([L[0].i, L[1].i], axis=1)
        o = np.lexsort((i, d), axis=1)[:, :K]
        ll = live[c]
        out[I[c][ll]] = np.take_along_axis(i, o, axis=1)[ll]
    return rounds, out


def main():
    from glim_amd import synth
    from oracle import oracle as orc

    pts = np.asarray(synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(64, 1024), 0))[:, :3].astype(np.float32)
    ref = orc.knn(pts.astype(np.float64), K)
    for sel in (None, (8, 12), (8, 6)):
        rounds, out = run(pts, sel)
        live = rounds[rounds > 0]
        print(f"{len(pts)} pts, select {sel}: rounds per wavefront mean {live.mean():.1f} p99 {np.percentile(live, 99):.0f} max {live.max()}; exact {bool((out == ref).all())}", flush=True)


if __name__ == "__main__":
    main()
