"""CPU model of the Hilbert-chunk kNN kernel (glim_amd/csrc/knn_chunks.hip, knn_chunk_kernel): reproduces the per-wavefront work counters the
kernel dumps with the diagnostic switch knn_debug=<file> (GLIM_AMD_DIAG) -- chunks scanned ("tiles") and lock-step insertion rounds -- for a given cloud, so that changes of
the visiting order / seeding can be evaluated without a GPU (the counters explain 83 % of the wavefront times: tools/knn_debug.py).
Not part of the product and not an oracle: a design tool.

  python tools/knn_model.py [lidar|rgbd] [measured.npy from tools/knn_debug.py]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHUNK, K = 64, 10


def hilbert_keys(pts, bits=13):
    """curve_key_kernel: quantise to `bits` per axis over the bounding box (largest extent), Skilling's transform, interleave."""
    p = pts.astype(np.float32)
    lo = p.min(axis=0)
    ext = np.float32((p.max(axis=0) - lo).max())
    qmax = np.uint32((1 << bits) - 1)
    scale = np.float32(qmax) / ext if ext > 0 else np.float32(0)
    X = [np.minimum(qmax, np.maximum(np.float32(0), (p[:, a] - lo[a]) * scale).astype(np.uint32)) for a in range(3)]
    M = np.uint32(1 << (bits - 1))
    Q = M
    while Q > 1:
        P = np.uint32(Q - 1)
        for a in range(3):
            hit = (X[a] & Q) != 0
            t = np.where(hit, np.uint32(0), (X[0] ^ X[a]) & P)
            X[0] = np.where(hit, X[0] ^ P, X[0] ^ t)
            if a != 0:
                X[a] = X[a] ^ t
        Q = np.uint32(Q >> 1)
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ np.uint32(Q - 1), t)
        Q = np.uint32(Q >> 1)
    X = [x ^ t for x in X]

    def spread3(v):
        out = np.zeros(len(v), dtype=np.uint64)
        v = v.astype(np.uint64)
        for b in range(bits):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out

    return spread3(X[2]) | (spread3(X[1]) << np.uint64(1)) | (spread3(X[0]) << np.uint64(2))


class Wave:
    """The 64 sorted top-K lists of one wavefront."""

    def __init__(self, self_idx):
        self.d = np.full((CHUNK, K), np.inf)
        self.i = np.repeat(self_idx[:, None], K, axis=1).astype(np.int64)

    def push(self, lanes, dn, idn):
        """TopK::push for the given lanes (exact (distance, index) order)."""
        if len(lanes) == 0:
            return
        d, i = self.d[lanes], self.i[lanes]
        ok = (dn < d[:, -1]) | ((dn == d[:, -1]) & (idn < i[:, -1]))
        if not ok.any():
            return
        lanes, d, i, dn, idn = lanes[ok], d[ok], i[ok], dn[ok], idn[ok]
        before = (d < dn[:, None]) | ((d == dn[:, None]) & (i < idn[:, None]))  # entries that order before the new one
        pos = before.sum(axis=1)
        cols = np.arange(K)[None, :]
        nd = np.where(cols < pos[:, None], d, np.where(cols == pos[:, None], dn[:, None], np.roll(d, 1, axis=1)))
        ni = np.where(cols < pos[:, None], i, np.where(cols == pos[:, None], idn[:, None], np.roll(i, 1, axis=1)))
        self.d[lanes], self.i[lanes] = nd, ni


def sqd(q, c):
    d = q[:, None, :] - c[None, :, :]
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def run(pts, order="index", seeds=K + 2, verbose=False, seed_mode="wrap", levels=1, level_scans=3, factor=0.5, margin=1.5, popmin=3, bands=None, heavy=None, select_slack=None, select_bits=None):
    n = len(pts)
    keys = hilbert_keys(pts)
    perm = np.argsort(keys, kind="stable")
    C = (n + CHUNK - 1) // CHUNK
    spt = np.full((C * CHUNK, 3), np.inf)
    spt[:n] = pts[perm].astype(np.float64)
    sidx = np.full(C * CHUNK, -1, dtype=np.int64)
    sidx[:n] = perm
    spt3 = spt.reshape(C, CHUNK, 3)
    sidx2 = sidx.reshape(C, CHUNK)
    live = sidx2 >= 0
    lo = np.where(live[..., None], spt3, np.inf).min(axis=1).astype(np.float32)
    hi = np.where(live[..., None], spt3, -np.inf).max(axis=1).astype(np.float32)
    G = (C + CHUNK - 1) // CHUNK
    tiles = np.zeros(C, dtype=np.int64)
    run.selections = np.zeros(C, dtype=np.int64)  # scans that ran the threshold selection (select_bits)
    rounds = np.zeros(C, dtype=np.int64)
    out = np.zeros((n, K), dtype=np.int64)
    all_lanes = np.arange(CHUNK)

    for c in range(C):
        q = np.where(live[c][:, None], spt3[c], spt3[c][0])  # padding lanes query the chunk's first point
        w = Wave(np.where(live[c], sidx2[c], -1))

        def scan(cc, need, seed):
            tiles[c] += 1
            cand, cidx = spt3[cc], sidx2[cc]
            seeded = np.zeros((CHUNK, CHUNK), dtype=bool)
            if seed:
                for t in range(seeds):
                    off = ((t + 1) >> 1) if (t & 1) else -(t >> 1)
                    if seed_mode == "wrap":  # the kernel: +-1, +-2, ... modulo the chunk
                        j = (all_lanes + off) & (CHUNK - 1)
                    else:  # a window of `seeds` curve neighbours that stays inside the chunk
                        j = np.clip(all_lanes - seeds // 2, 0, CHUNK - seeds) + t
                    seeded[all_lanes, j] = True
                    okl = all_lanes[cidx[j] >= 0]
                    dj = ((q[okl] - cand[j[okl]]) ** 2)
                    w.push(okl, (dj[:, 0] + dj[:, 1]) + dj[:, 2], cidx[j[okl]])
            thr = np.where(need, w.d[:, -1], -1.0)
            dall = sqd(q, cand)
            # FP32 mask pass: inflated bound (the few extra accepts of the FP32 evaluation are pops too)
            d32 = dall.astype(np.float32)
            real = (cidx >= 0)[None, :]
            m = (d32 <= (thr * 1.000002).astype(np.float32)[:, None] + np.float32(1e-37)) & ~seeded
            if (isinstance(levels, tuple) or levels > 1) and tiles[c] <= level_scans:
                # threshold cascade: the tightest level t_l = thr * factor^l for which (list entries <= t_l) + (new candidates <= t_l) >= K;
                # candidates beyond it cannot enter the list once those are inserted
                facs = levels if isinstance(levels, tuple) else tuple(factor ** l for l in range(levels))
                for l in range(len(facs) - 1, 0, -1):
                    tl = thr * facs[l]
                    tl32 = (tl * 1.000002).astype(np.float32)[:, None] + np.float32(1e-37)
                    ml = (d32 <= tl32) & ~seeded & real
                    have = (w.d <= tl[:, None]).sum(axis=1) + ml.sum(axis=1)
                    take = (have >= K) & need
                    upd = take & (ml.sum(axis=1) < m.sum(axis=1))
                    m[upd] = ml[upd] | (m[upd] & ~real)  # padding candidates stay (they are popped and skipped, as in the kernel)
            if not isinstance(levels, tuple) and levels < 0 and tiles[c] <= level_scans:
                # adaptive cascade: -levels thresholds spaced geometrically between t1 and thr, t1 from the lane's own count at thr under the
                # assumption "entries within t grow like t":  T(thr) = (list entries) + (accepted)  ->  t1 = thr * margin * K / T(thr)
                n0 = (m & real).sum(axis=1)
                have0 = np.isfinite(w.d).sum(axis=1)
                with np.errstate(invalid="ignore", divide="ignore"):
                    f1 = np.minimum(1.0, margin * K / np.maximum(have0 + n0, 1))
                L = -levels
                refine = need & (n0 > popmin) & np.isfinite(thr)
                best_m = m.copy()
                done = np.zeros(CHUNK, bool)
                for l in range(L):  # tightest first
                    fl = f1 ** ((L - l) / L)
                    tl = np.where(np.isfinite(thr), thr * fl, np.inf)
                    tl32 = (tl * 1.000002).astype(np.float32)[:, None] + np.float32(1e-37)
                    ml = (d32 <= tl32) & ~seeded & real
                    have = (w.d <= tl[:, None]).sum(axis=1) + ml.sum(axis=1)
                    take = refine & ~done & (have >= K)
                    best_m[take] = ml[take] | (m[take] & ~real)
                    done |= take
                m = best_m
            if select_bits is not None and m.sum(axis=1).max() > select_bits[1]:
                # the SELECT code of knn_chunks.hip, step by step: bisection over FP32 bit patterns (select_bits[0] steps, 16 octaves below the
                # bound), counting list entries (FP64) and accepted candidates (FP32); candidates beyond keep = t * 1.000002f + 1e-37f are dropped
                f32 = np.float32
                run.selections[c] += 1
                dv = np.where(m, d32, f32(np.inf))
                thr32 = ((thr * 1.000002).astype(f32) + f32(1e-37)).astype(f32)

                def count_le(t):
                    return (dv <= t[:, None]).sum(axis=1) + (w.d <= t.astype(np.float64)[:, None]).sum(axis=1)

                hi = np.minimum(thr32, f32(3.4028234e38)).view(np.uint32).astype(np.int64)
                hi = np.where(thr32 < 0, 0, hi)  # lanes that do not need the chunk (thr = -1): never selected
                sel = need & (count_le(hi.astype(np.uint32).view(f32)) >= K)
                lo = np.where(hi > (16 << 23), hi - (16 << 23), 0)
                for _ in range(select_bits[0]):
                    mid = lo + ((hi - lo) >> 1)
                    ok = count_le(mid.astype(np.uint32).view(f32)) >= K
                    hi = np.where(ok, mid, hi)
                    lo = np.where(ok, lo, mid + 1)
                keep = (hi.astype(np.uint32).view(f32) * f32(1.000002) + f32(1e-37)).astype(f32)
                km = dv <= keep[:, None]
                m[sel] &= km[sel]
            if select_slack is not None and tiles[c] <= level_scans:
                # per-lane threshold selection by bisection on the FP32 distances held in registers: the K-th of (list + accepted) found to a relative
                # slack; everything beyond it is dropped without being popped
                dm = np.where(m & real, dall, np.inf)
                kth = np.sort(np.concatenate([w.d, dm], axis=1), axis=1)[:, K - 1]
                m = m & ((d32 <= (kth * (1.0 + select_slack) * 1.000002).astype(np.float32)[:, None] + np.float32(1e-37)) | ~real)
            if not isinstance(levels, tuple) and levels == 0 and tiles[c] <= level_scans:
                # lower bound of any thresholding scheme: pop exactly the candidates that are in the list after this scan
                dm = np.where(m & real, dall, np.inf)
                alld = np.concatenate([w.d, dm], axis=1)
                kth = np.sort(alld, axis=1)[:, K - 1]
                m = m & ((dall <= kth[:, None]) | ~real)
            if bands is not None and tiles[c] <= level_scans:
                # banded pops (wave-level): thresholds thr * bands[l] (bands[0] = 1 > bands[1] > ...); the bands are popped from the tightest outwards,
                # each in its own lock-step loop; a lane whose K-th best is within the band's threshold after the band is done drops the rest
                alive = need.copy()
                prev = np.zeros_like(m)
                for l in range(len(bands) - 1, -1, -1):
                    tl = np.where(np.isfinite(thr), thr * bands[l], np.inf) if l > 0 else thr
                    tl32 = (tl * 1.000002).astype(np.float32)[:, None] + np.float32(1e-37)
                    ml = (d32 <= tl32) & m
                    band = ml & ~prev & alive[:, None]
                    prev = ml
                    cntb = band.sum(axis=1)
                    rb = int(cntb.max())
                    rounds[c] += rb
                    if rb:
                        jj = np.argsort(~band, axis=1, kind="stable")
                        for t in range(rb):
                            lanes = all_lanes[cntb > t]
                            j = jj[lanes, t]
                            good = cidx[j] >= 0
                            w.push(lanes[good], dall[lanes[good], j[good]], cidx[j[good]])
                    alive &= ~(w.d[:, -1] <= tl)
                return
            cnt = m.sum(axis=1)
            r = int(cnt.max()) if len(cnt) else 0
            if heavy is not None:
                # heavy lanes (more than heavy[0] accepted candidates) are merged cooperatively by the whole wavefront (rank by counting), at the
                # price of heavy[1] insertion rounds each; the others pop in lock step as before
                hv = cnt > heavy[0]
                rounds[c] += int(cnt[~hv].max() if (~hv).any() else 0) + int(np.ceil(heavy[1] * hv.sum()))
                rounds[c] -= r  # (added back below)
            rounds[c] += r
            if r == 0:
                return
            # lanes pop their masks in candidate order, one per round
            jj = np.argsort(~m, axis=1, kind="stable")  # accepted candidate positions first, in index order
            for t in range(r):
                lanes = all_lanes[cnt > t]
                j = jj[lanes, t]
                good = cidx[j] >= 0
                w.push(lanes[good], dall[lanes[good], j[good]], cidx[j[good]])

        scan(c, np.ones(CHUNK, bool), True)
        if c > 0:
            scan(c - 1, np.ones(CHUNK, bool), False)
        if c + 1 < C:
            scan(c + 1, np.ones(CHUNK, bool), False)

        def need_of(cc):
            g = np.maximum(0.0, np.maximum(lo[cc].astype(np.float64) - q, q - hi[cc].astype(np.float64)))
            return ((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]) * (1.0 - 1e-12) <= w.d[:, -1]

        def coarse(cands):
            r2 = w.d[:, -1].max()
            R = np.float32(np.sqrt(r2) * 1.000001) + np.float32(1e-30)
            ok = np.ones(len(cands), bool)
            for a in range(3):
                ok &= (lo[cands, a] <= hi[c, a] + R) & (hi[cands, a] >= lo[c, a] - R)
            return cands[ok]

        if order == "index":  # the kernel: groups of 64 chunks from the own group outwards, candidates of a group in index order
            gc = c // CHUNK
            for t in range(2 * G):
                gi = gc + ((t + 1) >> 1) if (t & 1) else gc - (t >> 1)
                if gi < 0 or gi >= G:
                    continue
                cands = np.arange(gi * CHUNK, min((gi + 1) * CHUNK, C))
                cands = cands[(cands != c) & (cands != c - 1) & (cands != c + 1)]
                for cc in coarse(cands):
                    need = need_of(cc)
                    if need.any():
                        scan(cc, need, False)
        elif order == "group_nearest":  # same groups, candidates of a group nearest box first
            gc = c // CHUNK
            for t in range(2 * G):
                gi = gc + ((t + 1) >> 1) if (t & 1) else gc - (t >> 1)
                if gi < 0 or gi >= G:
                    continue
                cands = np.arange(gi * CHUNK, min((gi + 1) * CHUNK, C))
                cands = coarse(cands[(cands != c) & (cands != c - 1) & (cands != c + 1)])
                gap = np.maximum(0, np.maximum(lo[cands] - hi[c], lo[c] - hi[cands])).astype(np.float64)
                for cc in cands[np.argsort((gap ** 2).sum(axis=1), kind="stable")]:
                    need = need_of(cc)
                    if need.any():
                        scan(cc, need, False)
        elif order == "global_nearest":  # one coarse test of ALL chunks with the radius after the first three scans, nearest box first
            cands = np.arange(C)
            cands = coarse(cands[(cands != c) & (cands != c - 1) & (cands != c + 1)])
            gap = np.maximum(0, np.maximum(lo[cands] - hi[c], lo[c] - hi[cands])).astype(np.float64)
            tiles_listed = len(cands)
            for cc in cands[np.argsort((gap ** 2).sum(axis=1), kind="stable")]:
                need = need_of(cc)
                if need.any():
                    scan(cc, need, False)
        ll = live[c]
        out[sidx2[c][ll]] = w.i[ll]
        if verbose and c % 256 == 0:
            print(f"  chunk {c}/{C}", flush=True)
    return tiles, rounds, out


def main():
    from glim_amd import synth

    which = sys.argv[1] if len(sys.argv) > 1 else "lidar"
    if which == "lidar":
        pts = synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(128, 1024), 0)
    else:
        pts = synth.scan(synth.Scene.small_room(), synth.pose(-2.5, -1.5, 1.4, 0.5), synth.pinhole_directions(640, 480, 70, 55), 0, sigma=0.002, max_range=8.0, min_range=0.3)
    pts = np.asarray(pts)[:, :3].astype(np.float32)
    measured = np.load(sys.argv[2]) if len(sys.argv) > 2 else None
    ref = None
    # what was evaluated in round 2 (profiles/r02/probe/knn_model_results.txt); the first entry is the kernel as shipped
    configs = [("index", {}),
               ("global_nearest", {}),                                                                   # one coarse test of all chunks, nearest box first
               ("index", {"seed_mode": "window"}),                                                       # seeds that do not wrap around the chunk
               ("index", {"levels": (1, 0.6, 0.3, 0.1, 0.01), "level_scans": 1000}),                     # counting cascade of thresholds
               ("index", {"levels": (1, 0.8, 0.6, 0.4, 0.2, 0.1, 0.03, 0.003), "level_scans": 1000}),
               ("index", {"bands": (1, 0.5, 0.25, 0.06), "level_scans": 1000}),                          # banded pops
               ("index", {"heavy": (16, 2.5)}),                                                          # cooperative merge of heavy lanes
               ("index", {"select_bits": (8, 6)}),                                                      # the SELECT code of knn_chunks.hip, emulated step by step
               ("index", {"select_bits": (8, 12)}),
               ("index", {"select_bits": (5, 6)}),
               ("index", {"select_slack": 0.04, "level_scans": 3}),                                      # bisection to 4 % in the first three scans only
               ("index", {"select_slack": 0.04, "level_scans": 1000}),                                   # ... in every scan
               ("global_nearest", {"select_slack": 0.04, "level_scans": 1000}),
               ("index", {"levels": 0, "level_scans": 1000}),                                            # lower bound of any thresholding scheme
               ("global_nearest", {"levels": 0, "level_scans": 1000})]
    for order, kw in configs:
        t0 = time.time()
        tiles, rounds, out = run(pts, order, **kw)
        order = order + " " + str(kw)
        # the model's time estimate uses the fit of tools/knn_debug.py on the measured wavefronts: 47.2 + 2.68 tiles + 0.509 rounds (us)
        est = 47.2 + 2.68 * tiles + 0.509 * rounds
        print(f"{order}: tiles mean {tiles.mean():.1f} max {tiles.max()}  rounds mean {rounds.mean():.1f} p99 {np.percentile(rounds, 99):.0f} max {rounds.max()}  "
              f"modelled wavefront us mean {est.mean():.0f} max {est.max():.0f}   ({time.time() - t0:.0f}s)", flush=True)
        if ref is None:
            ref = out
            if measured is not None:
                print("  vs measured counters: tiles equal %.3f, rounds equal %.3f, mean |d rounds| %.2f" %
                      ((measured[:, 0] == tiles).mean(), (measured[:, 1] == rounds).mean(), np.abs(measured[:, 1] - rounds).mean()))
        else:
            print("  lists identical to the index order:", bool((out == ref).all()))


if __name__ == "__main__":
    main()
