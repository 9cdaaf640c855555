import os, sys, time, numpy as np
sys.path.insert(0, '.')
from glim_amd import api, synth
from oracle import oracle as orc
CHECK = os.environ.get('KNN_TIME_NOCHECK') is None  # timing only: skip the oracle comparison


def ref_knn(pts):  # the oracle's lists, cached across the processes of one A/B session (same seeded points every time)
    path = '/tmp/knn_ref_%d.npy' % len(pts)
    if os.path.exists(path):
        return np.load(path)
    ref = orc.knn(pts, 10)
    np.save(path, ref)
    return ref


ctx = api.Context(0, 1)
scene = synth.Scene.default()
for rings, az in ((128, 1024), (64, 1024)):
    pts = synth.scan(scene, synth.arc_trajectory(1)[0], synth.lidar_directions(rings, az), 0)
    g = api.PointCloudGPU.clone(pts, ctx=ctx)
    g.find_neighbors(10, download=False)
    t = time.perf_counter(); 
    for _ in range(5): g.find_neighbors(10, download=False)
    dt = (time.perf_counter() - t) / 5
    nb = g.find_neighbors(10)
    print(len(pts), 'knn ms', dt * 1e3, 'exact', bool((nb == ref_knn(pts)).all()) if CHECK else 'unchecked', flush=True)
    t = time.perf_counter(); g.estimate_covariances(10); print(' cov ms', (time.perf_counter() - t) * 1e3)
    t = time.perf_counter(); vm = api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(g); print(' vmap ms', (time.perf_counter() - t) * 1e3)
# dense depth camera frame (config 5)
room = synth.Scene.small_room()
dirs = synth.pinhole_directions(640, 480, 70, 55)
pts = synth.scan(room, synth.pose(-2.5, -1.5, 1.4, 0.5), dirs, 0, sigma=0.002, max_range=8.0, min_range=0.3)
g = api.PointCloudGPU.clone(pts, ctx=ctx)
g.find_neighbors(10, download=False)
t = time.perf_counter()
for _ in range(5): g.find_neighbors(10, download=False)
dt = (time.perf_counter() - t) / 5
nb = g.find_neighbors(10)
print(len(pts), 'rgbd knn ms', dt * 1e3, 'exact', bool((nb == ref_knn(pts)).all()) if CHECK else 'unchecked', flush=True)
