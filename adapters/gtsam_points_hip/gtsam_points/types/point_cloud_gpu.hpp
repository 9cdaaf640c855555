// gtsam_points/types/point_cloud_gpu.hpp, MI355X edition: gtsam_points::PointCloudGPU::clone(frame[, stream])
// (odometry_estimation_gpu.cpp:96; sub_mapping.cpp:168,393; global_mapping.cpp:253,260,743).
// Like upstream it is a PointCloudCPU (the host copy GLIM keeps reading) that also owns the device image; `points_gpu` is set to a non-null
// tag because GLIM uses it only as a "lives on the GPU" flag -- the device layout is the library's own SoA (glim_amd.h), not Vector3f arrays.
#pragma once

#include <memory>

#include <gtsam_points/types/point_cloud_cpu.hpp>

#include <glim_amd_gtsam.hpp>
#include <gtsam_points/cuda/cuda_stream.hpp>

struct CUstream_st;

namespace gtsam_points {

class PointCloudGPU : public PointCloudCPU {
public:
  using Ptr = std::shared_ptr<PointCloudGPU>;
  using ConstPtr = std::shared_ptr<const PointCloudGPU>;

  // deep copy of the host attributes + upload (Vector4d points / Matrix4d covariances / Vector4d normals -> device)
  // stream: the module's CUDAStream, i.e. the context whose stream and mutex the upload uses
  static Ptr clone(const PointCloud& frame, CUstream_st* stream = nullptr) {
    Ptr out(new PointCloudGPU());
    const PointCloudCPU::Ptr host = PointCloudCPU::clone(frame);
    static_cast<PointCloudCPU&>(*out) = *host;  // upstream keeps the CPU attributes alongside the device ones
    out->device_ = glim_amd::clone(frame, glim_amd::context_of(stream));
    out->points_gpu = reinterpret_cast<decltype(out->points_gpu)>(out->device_->handle());
    return out;
  }
  const glim_amd::PointCloudGPU::ConstPtr& device() const { return device_; }
  size_t memory_usage_gpu() const { return device_ ? device_->memory_usage_gpu() : 0; }

private:
  PointCloudGPU() {}
  glim_amd::PointCloudGPU::ConstPtr device_;
};

// the device image of a frame a GLIM call site passes as gtsam_points::PointCloud::ConstPtr; frames that were not made by PointCloudGPU::clone
// are uploaded on the spot (upstream aborts with "source points_gpu is null" instead)
inline glim_amd::PointCloudGPU::ConstPtr device_cloud(const PointCloud::ConstPtr& frame) {
  if (auto gpu = std::dynamic_pointer_cast<const PointCloudGPU>(frame)) return gpu->device();
  return glim_amd::clone(*frame);
}

}  // namespace gtsam_points
