// stand-in for gtsam_points/util/parallelism.hpp: the OpenMP backend is the default one
#pragma once
namespace gtsam_points {
inline bool is_omp_default() { return true; }
inline bool is_tbb_default() { return false; }
}  // namespace gtsam_points
