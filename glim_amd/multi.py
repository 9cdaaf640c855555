"""Multi-GPU evaluation of a multi-scan VGICP cost: factor-list sharding + all-reduce of per-factor blocks.

The reference is single-device (no NCCL/MPI anywhere); this is the MI355X-native extension BASELINE.json's
north_star asks for.  One process per GPU (`torch.distributed`; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
Clouds and voxel maps are replicated on every GPU (they are tiny next to 288 GB of HBM); only the FACTOR LIST is sharded, so no
point data ever crosses xGMI.  Each rank writes the compact 29-double record of its own factors into its rows of a dense
[n_factors x 29] FP64 array (all other rows zero) and ONE all-reduce(sum) completes the cost on every rank.  The binary factor's
target-side blocks are expanded from the compact source block after the reduction (glim_amd_expand_compact), which keeps the
message at 232 B per factor instead of 976 B.
"""
import numpy as np

COMPACT = 29


def shard_bounds(costs, world_size):
    """Contiguous, cost-balanced split of the factor list.  costs[i] ~ number of source points of factor i.
    Returns world_size + 1 boundaries; rank r owns factors [b[r], b[r+1])."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    if n == 0:
        return [0] * (world_size + 1)
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        b = int(np.searchsorted(cum, target, side="left"))
        # choose the nearer boundary, keep monotone
        if b > 0 and abs(cum[b - 1] - target) <= abs(cum[min(b, n)] - target):
            b -= 1
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return bounds


def my_range(costs, rank, world_size):
    b = shard_bounds(costs, world_size)
    return b[rank], b[rank + 1]


def allreduce_blocks(blocks, group=None):
    """In-place sum over ranks of the [n_factors x 29] array (torch tensor on the device for nccl/RCCL, on the host for gloo)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(blocks, op=dist.ReduceOp.SUM, group=group)
    return blocks


class ShardedCostEvaluator:
    """Evaluates H/b/error for a list of (target voxel map, source cloud, flags) factors across ranks.

    `make_local_set(lo, hi)` must return an object with `linearize_device_async(poses[lo:hi], out_ptr, row_offset)` (the HIP
    NonlinearFactorSetGPU) -- or, in the CPU tests, any callable stand-in with `linearize_rows(poses) -> [hi-lo x 29]`.
    """

    def __init__(self, costs, rank, world_size):
        self.costs = list(costs)
        self.rank, self.world_size = rank, world_size
        self.lo, self.hi = my_range(self.costs, rank, world_size)
        self.n = len(self.costs)

    def owned(self):
        return range(self.lo, self.hi)

    def evaluate_host(self, local_rows):
        """CPU/gloo form: local_rows is [(hi-lo) x 29]; returns the reduced [n x 29] array on every rank."""
        import torch

        blocks = torch.zeros(self.n, COMPACT, dtype=torch.float64)
        if self.hi > self.lo:
            blocks[self.lo:self.hi] = torch.as_tensor(np.asarray(local_rows, dtype=np.float64).reshape(self.hi - self.lo, COMPACT))
        return allreduce_blocks(blocks)

    def evaluate_device(self, fset, poses, blocks):
        """GPU/RCCL form: `fset` holds this rank's factors, `blocks` is a zeroed [n x 29] float64 CUDA tensor."""
        if self.hi > self.lo:
            fset.linearize_device_async(poses[self.lo:self.hi], blocks.data_ptr(), self.lo)
        return allreduce_blocks(blocks)


def compact_from_linearized(L):
    """[num_inliers, error, 21 upper-triangular H_ss entries, 6 b_s] from a linearised-factor dict (layout of the device record)."""
    iu = np.triu_indices(6)
    return np.concatenate([[float(L["num_inliers"]), float(L["error"])], np.asarray(L["H_ss"])[iu], np.asarray(L["b_s"])])
