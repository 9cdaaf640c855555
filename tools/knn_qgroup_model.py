"""CPU model of the query-group kNN kernel (glim_amd/csrc/knn_qgroup.hip, knn_qgroup_kernel): the same walk, the same conservative FP32 tests
(box gaps deflated by 0.99999 against the bound's FP32 image inflated by 1.00001; the FP32 distance image that decides whether a chunk is evaluated
exactly at all), the same exact FP64 `(dx^2 + dy^2) + dz^2`, the same acceptance order -- lane by lane, the ballot of the remaining candidates
re-taken against the tightened bound after every insertion -- for a given cloud, so that the pruning argument of the kernel can be checked against
brute force without a GPU, and its work counters (chunk scans, exact evaluations, insertions per wavefront) predicted.
Not part of the product and not an oracle: a design / test tool.  The bitonic network of the kernel is modelled by a sort on (distance, index):
both produce the unique ascending order of distinct keys.

  python tools/knn_qgroup_model.py            # a 4 096-point scan subset: lists against brute force + counters
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knn_model import hilbert_keys  # noqa: E402  (tools/knn_model.py: curve_key_kernel's keys)

CHUNK = 64
F32 = np.float32
INF32 = F32(np.inf)


def _fma32(a, b, c):
    """fmaf on float32 arrays: a * b is exact in float64 (24 x 24 bits); the sum is rounded to float64 and then to float32 (a double rounding that
    differs from a true fma in rare last-place cases -- both stay far inside the 1e-5 margin the kernel's test allows for)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def build(pts, bits):
    """knn_curve's preparation: curve order (stable by index), chunks of 64 with padding, chunk boxes, boxes of the groups of 64 chunks."""
    n = len(pts)
    order = np.argsort(hilbert_keys(pts, bits), kind="stable")
    C = (n + CHUNK - 1) // CHUNK
    sp = np.zeros((C * CHUNK, 3), dtype=F32)
    si = np.full(C * CHUNK, -1, dtype=np.int64)
    sp[:n] = pts[order]
    si[:n] = order
    box = np.zeros((C, 6), dtype=F32)
    for c in range(C):
        v = sp[c * CHUNK:(c + 1) * CHUNK][si[c * CHUNK:(c + 1) * CHUNK] >= 0]
        box[c, :3], box[c, 3:] = v.min(axis=0), v.max(axis=0)
    G = (C + CHUNK - 1) // CHUNK
    gbox = np.zeros((G, 6), dtype=F32)
    for g in range(G):
        b = box[g * CHUNK:(g + 1) * CHUNK]
        gbox[g, :3], gbox[g, 3:] = b[:, :3].min(axis=0), b[:, 3:].max(axis=0)
    return sp, si, box, gbox, C, G


def run(pts, k=10, queries_per_wave=2, bits=None):
    """Returns (lists [n, k], counters per wavefront: scans, exact query-chunk evaluations, insertions, chunk-test rounds)."""
    pts = np.asarray(pts, dtype=F32)
    n = len(pts)
    K = next(x for x in (8, 10, 16, 24, 32) if k <= x)  # DISPATCH_K
    Q = queries_per_wave
    sp, si, box, gbox, C, G = build(pts, bits if bits is not None else (8 if n < 32768 else 13))
    out = np.zeros((n, k), dtype=np.int64)
    counters = []
    for c in range(C):
        for sub in range(CHUNK // Q):
            base = c * CHUNK + sub * Q
            self_idx = si[base:base + Q]
            if self_idx[0] < 0:
                continue
            q32 = sp[base:base + Q].copy()
            q32[self_idx < 0] = q32[0]  # padding queries repeat the group's first query
            q64 = q32.astype(np.float64)
            ld = np.full((Q, K), np.inf)
            li = np.full((Q, K), 0x7FFFFFFF, dtype=np.int64)
            bd, bi = np.full(Q, np.inf), np.full(Q, 0x7FFFFFFF, dtype=np.int64)
            bd32 = np.full(Q, INF32)
            cnt = [0, 0, 0, 0]

            def refresh(i):
                bd[i], bi[i] = ld[i, K - 1], li[i, K - 1]
                with np.errstate(over="ignore"):
                    bd32[i] = F32(F32(bd[i]) * F32(1.00001) + F32(1e-37))

            def scan(cc, first):
                p = sp[cc * CHUNK:(cc + 1) * CHUNK]
                raw = si[cc * CHUNK:(cc + 1) * CHUNK]
                valid = raw >= 0
                cidx = np.where(valid, raw, 0x7FFFFFFF)
                cnt[0] += 1
                for i in range(Q):
                    if not first:
                        dx, dy, dz = q32[i, 0] - p[:, 0], q32[i, 1] - p[:, 1], q32[i, 2] - p[:, 2]
                        d32 = _fma32(dz, dz, _fma32(dy, dy, (dx * dx).astype(F32)))
                        if not (valid & (d32 <= bd32[i])).any():
                            continue
                    cnt[1] += 1
                    e = q64[i][None, :] - p.astype(np.float64)
                    d = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
                    d = np.where(valid, d, np.inf)
                    if first:
                        o = np.lexsort((cidx, d))  # the 64-lane bitonic network: ascending (d, idx)
                        ld[i], li[i] = d[o[:K]], cidx[o[:K]]
                        refresh(i)
                        continue
                    m = (d < bd[i]) | ((d == bd[i]) & (cidx < bi[i]))
                    while m.any():
                        j = int(np.argmax(m))  # ctz of the ballot
                        dc, ic = d[j], cidx[j]
                        pos = int(((ld[i] < dc) | ((ld[i] == dc) & (li[i] < ic))).sum())
                        ld[i, pos + 1:], li[i, pos + 1:] = ld[i, pos:K - 1].copy(), li[i, pos:K - 1].copy()
                        ld[i, pos], li[i, pos] = dc, ic
                        refresh(i)
                        cnt[2] += 1
                        m[j] = False
                        m &= (d < bd[i]) | ((d == bd[i]) & (cidx < bi[i]))

            def gaps(b):  # b: [m, 6] boxes -> [m, Q] deflated squared gaps (FP32)
                g2 = np.zeros((len(b), Q), dtype=F32)
                for i in range(Q):
                    g = [np.maximum(F32(0), np.maximum(b[:, a] - q32[i, a], q32[i, a] - b[:, 3 + a])).astype(F32) for a in range(3)]
                    with np.errstate(over="ignore"):
                        g2[:, i] = ((g[0] * g[0] + g[1] * g[1]).astype(F32) + g[2] * g[2]).astype(F32) * F32(0.99999)
                return g2

            scan(c, True)
            if c > 0:
                scan(c - 1, False)
            if c + 1 < C:
                scan(c + 1, False)
            gc = c // CHUNK
            ok_groups = [g for g in range(G) if (gaps(gbox[g:g + 1])[0] <= bd32).any()] if G <= 256 else list(range(G))
            ok_groups.sort(key=lambda g: (abs(g - gc), g < gc))  # nearest first, the upper one on ties
            for g in ok_groups:
                lanes = np.arange(g * CHUNK, min(C, (g + 1) * CHUNK))
                mine = (lanes != c) & (lanes != c - 1) & (lanes != c + 1)
                g2 = gaps(box[lanes])
                todo = mine.copy()
                while todo.any():
                    cnt[3] += 1
                    todo &= mine & (g2 <= bd32[None, :]).any(axis=1)
                    if not todo.any():
                        break
                    j = int(np.argmax(todo))
                    todo[j] = False
                    scan(int(lanes[j]), False)
            for i in range(Q):
                if self_idx[i] >= 0:
                    out[self_idx[i]] = li[i, :k]
            counters.append(cnt)
    return out, np.array(counters)


def brute(pts, k):
    p = np.asarray(pts, dtype=F32).astype(np.float64)
    out = np.zeros((len(p), k), dtype=np.int64)
    idx = np.arange(len(p))
    for i in range(len(p)):
        e = p[i] - p
        d = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
        out[i] = np.lexsort((idx, d))[:k]
    return out


if __name__ == "__main__":
    from glim_amd import synth

    scan = synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(64, 512), 0)[:, :3]
    pts = scan[np.sort(np.random.default_rng(0).choice(len(scan), 4096, replace=False))].astype(F32)
    for q in (1, 2):
        got, cnt = run(pts, 10, q)
        print(f"{q} queries per wavefront: lists equal brute force: {bool((got == brute(pts, 10)).all())}; per wavefront: chunk scans {cnt[:, 0].mean():.1f}, "
              f"exact evaluations {cnt[:, 1].mean():.1f}, insertions {cnt[:, 2].mean():.1f}, chunk-test rounds {cnt[:, 3].mean():.1f}")
