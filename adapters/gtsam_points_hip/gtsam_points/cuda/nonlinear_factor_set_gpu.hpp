// gtsam_points/cuda/nonlinear_factor_set_gpu.hpp, MI355X edition: NonlinearFactorSetGPU::{add, linearize} (odometry_estimation_gpu.cpp:383-385)
// = glim_amd::NonlinearFactorSetHIP: every HIP factor of the graph in one fused launch.
#pragma once

#include <glim_amd_gtsam.hpp>

namespace gtsam_points {

class NonlinearFactorSetGPU : public glim_amd::NonlinearFactorSetHIP {
public:
  NonlinearFactorSetGPU() : glim_amd::NonlinearFactorSetHIP() {}
};

}  // namespace gtsam_points
