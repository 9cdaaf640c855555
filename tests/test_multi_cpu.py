"""N > 1 path on CPU: world_size-2 `gloo` processes shard a factor list, all-reduce the [n x 29] block array and expand it.

The per-factor compute is stood in by the CPU oracle here (tests may use it); on the GPU box the same ShardedCostEvaluator is
fed by the HIP NonlinearFactorSetGPU (bench.py --gpus N).  What this covers: shard boundaries, row placement, the collective,
the compact layout and the FP64 expansion of binary blocks -- i.e. everything of the distributed path that is not the kernel.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_are_contiguous_and_balanced():
    from glim_amd import multi

    rng = np.random.default_rng(0)
    for n, w in [(0, 2), (1, 2), (5, 8), (17, 2), (100, 8), (32640, 8)]:
        costs = rng.integers(1000, 70000, size=n)
        b = multi.shard_bounds(costs, w)
        assert b[0] == 0 and b[-1] == n and len(b) == w + 1
        assert all(b[i] <= b[i + 1] for i in range(w))
        if n >= 4 * w:
            loads = [costs[b[i]:b[i + 1]].sum() for i in range(w)]
            assert max(loads) <= 1.25 * (costs.sum() / w) + costs.max()
    assert multi.shard_bounds([1] * 8, 8) == list(range(9))


def test_gather_layout_places_every_factor_once():
    from glim_amd import multi

    rng = np.random.default_rng(1)
    for n, w in [(1, 2), (5, 8), (17, 2), (100, 8), (32640, 8), (32640, 3)]:
        costs = rng.integers(1000, 70000, size=n)
        b = multi.shard_bounds(costs, w)
        max_rows, index = multi.ShardedCostEvaluator(costs, 0, w).gather_layout()
        assert max_rows == max(b[r + 1] - b[r] for r in range(w)) and len(np.unique(index)) == n
        for r in range(w):
            assert np.array_equal(index[b[r]:b[r + 1]], r * max_rows + np.arange(b[r + 1] - b[r]))
        # a single rank gathers nothing: the assembled array is its own rows
        ev1 = multi.ShardedCostEvaluator(costs, 0, 1)
        rows = rng.normal(size=(n, multi.COMPACT))
        assert np.array_equal(ev1.gather_host(rows).numpy(), rows)


def test_world8_record_layout_against_the_unsharded_order():
    """What an 8-device node does with configs[3]'s 32 640 pairs, on the CPU: glim_amd_shard_bounds cuts the pair list, glim_amd_shard_layout
    (the very arithmetic glim_amd_multi_linearize, its ncclAllGather calls and glim_amd_multi_records use) places every factor's record.  A
    stand-in evaluation -- every "device" writes its own rows into ITS gathered array, the all-gather of each piece copies equal slots between
    them -- must reproduce the unsharded record array on every device, for every number of pieces."""
    from glim_amd import _lib, api

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    rng = np.random.default_rng(8)
    S = 256
    sizes = rng.integers(60000, 66000, size=S)
    pairs = [(i, j) for j in range(S) for i in range(j)]
    costs = np.array([sizes[j] for _, j in pairs], dtype=np.float64)
    n, world = len(pairs), 8
    truth = rng.normal(size=(n, 29))  # the records an unsharded evaluation would produce, factor order
    b = api.shard_bounds(costs, world)
    assert b[0] == 0 and b[-1] == n and all(b[r] < b[r + 1] for r in range(world))
    loads = [costs[b[r]:b[r + 1]].sum() for r in range(world)]
    assert max(loads) / np.mean(loads) < 1.005  # 4 080 pairs per shard: a boundary moves a load by one pair at most
    for split in (-1, 1, 2, 4, 8):
        rows, max_rows, pieces, piece_rows = api.shard_layout(b, split)
        assert max_rows == max(b[r + 1] - b[r] for r in range(world))
        assert pieces == (max(1, min(4, max_rows // 2048)) if split < 0 else split) and piece_rows == -(-max_rows // pieces)
        assert len(np.unique(rows)) == n and rows.min() >= 0 and rows.max() < world * max_rows
        gathered = [np.full((world * max_rows, 29), np.nan) for _ in range(world)]
        for d in range(world):  # every device's kernels write its own rows
            gathered[d][rows[b[d]:b[d + 1]]] = truth[b[d]:b[d + 1]]
        for p in range(pieces):  # ncclAllGather of piece p: region p is `world` equal slots, device d's at d * slot inside it
            rows_p = max(0, min(piece_rows, max_rows - p * piece_rows))
            start = world * p * piece_rows
            for d in range(world):
                lo = start + d * rows_p
                own = rows[b[d]:b[d + 1]]
                own_p = own[(own >= lo) & (own < lo + rows_p)]
                # the slot holds exactly this device's factors [p * piece_rows, (p + 1) * piece_rows) of its shard, in order
                k0 = p * piece_rows
                want = np.arange(min(max(0, (b[d + 1] - b[d]) - k0), rows_p)) + lo
                assert np.array_equal(own_p, want), (split, p, d)
                for e in range(world):
                    gathered[e][lo:lo + rows_p] = gathered[d][lo:lo + rows_p]
        for d in range(world):
            assert np.array_equal(gathered[d][rows], truth), (split, d)
    # ragged and empty shards (fewer factors than devices; a piece that some shards do not reach)
    for n2, w2 in [(3, 8), (100, 8), (1000, 3), (129, 2)]:
        c2 = rng.integers(1, 50, size=n2).astype(np.float64)
        b2 = api.shard_bounds(c2, w2)
        for split in (-1, 1, 2, 3):
            rows, max_rows, pieces, piece_rows = api.shard_layout(b2, split)
            assert len(np.unique(rows)) == n2 and (n2 == 0 or rows.max() < w2 * max_rows) and pieces * piece_rows >= max_rows > (pieces - 1) * piece_rows


def test_shard_bounds_properties():
    """Property test of glim_amd_shard_bounds (hypothesis): contiguous, covering, monotone; the heaviest shard exceeds the mean by at most one
    factor's cost (the boundary nearest to r / world of the cumulative cost is never further than half a factor away on either side)."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from glim_amd import _lib, api

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.integers(min_value=1, max_value=100000), min_size=0, max_size=400), st.integers(min_value=1, max_value=16))
    def check(costs, world):
        b = api.shard_bounds(costs, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == len(costs)
        assert all(b[r] <= b[r + 1] for r in range(world))
        if costs:
            c = np.asarray(costs, dtype=np.float64)
            loads = [c[b[r]:b[r + 1]].sum() for r in range(world)]
            assert sum(loads) == c.sum()
            assert max(loads) <= c.sum() / world + c.max() + 1e-9
        rows, max_rows, pieces, piece_rows = api.shard_layout(b, -1)
        assert len(np.unique(rows)) == len(costs)

    check()


def test_c_abi_shard_bounds_equals_python_rule():
    """The single-process multi-device entry (glim_amd_multi_set_factors) shards with glim_amd_shard_bounds; the one-process-per-GPU harness
    (glim_amd/multi.py) with shard_bounds: same rule, same boundaries."""
    from glim_amd import _lib, api, multi

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    rng = np.random.default_rng(3)
    for n, w in [(0, 1), (0, 4), (1, 1), (1, 8), (3, 8), (5, 2), (64, 8), (1000, 7), (32640, 8), (32640, 3)]:
        costs = rng.integers(1, 70000, size=n)
        assert api.shard_bounds(costs, w) == list(multi.shard_bounds(costs, w)), (n, w)
    assert api.shard_bounds([5, 5, 5, 5], 2) == [0, 2, 4]
    assert api.shard_bounds([100, 1, 1, 1], 2) == [0, 1, 4]
    with pytest.raises(api.GlimAmdError):
        api.shard_bounds([1.0], 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from glim_amd import api, multi, synth
    from oracle import oracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank builds the same replicated problem: 4 small scans, all ordered pairs as binary factors
        scene = synth.Scene.default()
        dirs = synth.lidar_directions(16, 64)
        poses = synth.arc_trajectory(4)
        scans, covs, maps = [], [], []
        for i, T in enumerate(poses):
            p = synth.scan(scene, T, dirs, i)
            _, c = orc.covariances(p, orc.knn(p, 10))
            c = c.astype(np.float32).astype(np.float64)
            scans.append(p)
            covs.append(c)
            maps.append(orc.VoxelMap(1.0).insert(p, c))
        pairs = [(i, j) for i in range(4) for j in range(4) if i != j]
        deltas = [synth.relative_pose(poses[i], poses[j]) for i, j in pairs]
        costs = [len(scans[j]) for _, j in pairs]
        ev = multi.ShardedCostEvaluator(costs, rank, world)
        rows = [multi.compact_from_linearized(orc.vgicp_linearize(maps[pairs[f][0]], scans[pairs[f][1]], covs[pairs[f][1]], deltas[f]))
                for f in ev.owned()]
        blocks = ev.evaluate_host(np.array(rows).reshape(-1, multi.COMPACT)).numpy()
        # the same exchange on rows the HIP kernels really produced (tests/golden/hip_rows.npz, captured on an MI355X by
        # tests/golden/make_golden_hip_rows.py for this very problem): shard, all-reduce, expand; the expansion must reproduce the
        # records the GPU run expanded itself, bit for bit, and agree with the oracle within the factor tolerance
        hip = dict(np.load(os.path.join(ROOT, "tests", "golden", "hip_rows.npz")))
        hip_blocks = ev.evaluate_host(hip["rows"][ev.lo:ev.hi]).numpy()
        assert np.array_equal(hip_blocks, hip["rows"])
        # the all-gather form of the exchange (owned rows only, shards padded to the longest) assembles the same array
        assert np.array_equal(ev.gather_host(hip["rows"][ev.lo:ev.hi]).numpy(), hip["rows"])
        assert np.array_equal(ev.gather_host(np.array(rows).reshape(-1, multi.COMPACT)).numpy(), blocks)
        hip_maps = [orc.VoxelMap(1.0).insert(p, c.astype(np.float64)) for p, c in zip(scans, hip["covs"])]
        for f, (i, j) in enumerate(pairs):
            got = api.expand_compact(hip_blocks[f], hip["deltas"][f], api.FACTOR_BINARY)
            assert np.array_equal(got["H_tt"], hip["H_tt"][f]) and np.array_equal(got["H_ts"], hip["H_ts"][f]) and np.array_equal(got["b_t"], hip["b_t"][f])
            D = np.eye(4)
            D[:3, :4] = hip["deltas"][f].reshape(3, 4)
            ref = orc.vgicp_linearize(hip_maps[i], scans[j], hip["covs"][j].astype(np.float64), D)
            assert got["num_inliers"] == ref["num_inliers"]
            assert np.abs(got["H_tt"] - ref["H_tt"]).max() <= 2e-4 * np.abs(ref["H_tt"]).max()
        # every rank must now hold every factor; compare with the unsharded evaluation
        worst = 0.0
        for f, (i, j) in enumerate(pairs):
            ref = orc.vgicp_linearize(maps[i], scans[j], covs[j], deltas[f])
            got = api.expand_compact(blocks[f], deltas[f], api.FACTOR_BINARY)
            assert got["num_inliers"] == ref["num_inliers"]
            for k in ("H_tt", "H_ss", "H_ts", "b_t", "b_s"):
                worst = max(worst, float(np.abs(got[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)))
        q.put((rank, ev.lo, ev.hi, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_equals_unsharded_gloo_world2():
    import torch.multiprocessing as mp

    from glim_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(420)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 12  # contiguous shards covering all 12 factors
    assert all(r[3] < 1e-9 for r in res)


def _ragged_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from glim_amd import multi

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        # (number of factors, costs): fewer factors than ranks (some rank owns nothing), one dominant factor, ragged shards
        for case, costs in enumerate([[7.0, 3.0], [1.0], [100.0, 1.0, 1.0, 1.0, 1.0], list(np.random.default_rng(5).integers(1, 9, size=23))]):
            n = len(costs)
            rows = np.random.default_rng(100 + case).normal(size=(n, multi.COMPACT))  # the same array on every rank
            ev = multi.ShardedCostEvaluator(costs, rank, world)
            ok = ok and np.array_equal(ev.gather_host(rows[ev.lo:ev.hi]).numpy(), rows)
            ok = ok and np.array_equal(ev.evaluate_host(rows[ev.lo:ev.hi]).numpy(), rows)
            # the two-halves form bench.py uses for N > 1 (gather of half A overlaps the kernels of half B): same method, host tensors, and
            # stand-in sets that write their rows where the HIP finalise kernel would
            import torch

            max_rows, h, index = ev.halves_layout()
            send = torch.zeros(max_rows, multi.COMPACT, dtype=torch.float64)
            gathered = torch.zeros(world * max_rows, multi.COMPACT, dtype=torch.float64)

            class FakeSet:
                def __init__(self, lo):
                    self.lo = lo

                def linearize_device_async(self, poses, ptr, row_offset):
                    assert ptr == send.data_ptr()
                    send[row_offset:row_offset + len(poses)] = torch.as_tensor(rows[self.lo:self.lo + len(poses)])

            (lo, mid), (_, hi) = ev.halves_ranges()
            ev.gather_device_halves(FakeSet(lo), FakeSet(mid), rows, send, gathered)  # `poses` only supplies the slice lengths here
            ok = ok and np.array_equal(gathered[torch.as_tensor(index)].numpy(), rows)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ragged_and_empty_shards_gloo_world3():
    """Three ranks, factor lists shorter than / not divisible by the world size: a rank that owns no factor still takes part in both forms of the
    exchange and every rank assembles the full array."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(420)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(3)) == [(0, True), (1, True), (2, True)]
