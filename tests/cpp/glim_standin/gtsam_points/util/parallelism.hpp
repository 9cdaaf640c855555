#pragma once
namespace gtsam_points {
bool is_omp_default();
bool is_tbb_default();
}  // namespace gtsam_points
