"""Device radix sort timing (debug entry point, includes the H2D/D2H of the host arrays -> use the rocprof kernel stats for kernels)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glim_amd import api
ctx = api.Context(0, 1)
rng = np.random.default_rng(0)
for n in (12000, 131072, 524288):
    keys = rng.integers(0, 2**32, size=n, dtype=np.uint64)
    for _ in range(3): api.debug_sort_pairs(keys, None, 32, ctx=ctx)
