// gtsam_points/cuda/nonlinear_factor_set_gpu_create.hpp, MI355X edition: the factory GLIM registers as a linearisation hook
// (src/glim/viewer/offline_viewer.cpp:28-30; in live runs glim_ros does it).
#pragma once

#include <memory>

#include <gtsam_points/cuda/nonlinear_factor_set_gpu.hpp>

namespace gtsam_points {

inline std::shared_ptr<NonlinearFactorSet> create_nonlinear_factor_set_gpu() { return std::make_shared<NonlinearFactorSetGPU>(); }

}  // namespace gtsam_points
