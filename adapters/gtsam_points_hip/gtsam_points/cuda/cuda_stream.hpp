// gtsam_points/cuda/cuda_stream.hpp, MI355X edition: gtsam_points::CUDAStream as GLIM uses it -- a RAII stream handed to PointCloudGPU::clone,
// voxel maps and overlap_gpu (src/glim/odometry/odometry_estimation_gpu.cpp:76, :96, :103, :231; sub_mapping.cpp:86; global_mapping.cpp:109).
// Here it OWNS a library context with one HIP stream (glim_amd_ctx_create_ex): every call that receives the stream runs on that context --
// its stream, its mutex -- so the three GLIM modules, which live in three threads (async_odometry_estimation.cpp:15, async_sub_mapping.cpp:8,
// async_global_mapping.cpp:24), never serialise on one another's uploads or map builds.  The CUstream_st* the reference passes around is the
// address of the context object; glim_amd::context_of() turns it back.  Stream priority: glim_amd::default_stream_priority() of the
// constructing thread (adapters/glim/odometry_estimation_hip_create.cpp raises it for the odometry).
#pragma once

#include <memory>

#include <glim_amd/gtsam_points_compat.hpp>

struct CUstream_st;  // the reference's stream handle type; never dereferenced as such

namespace glim_amd {
// the context behind a stream handle of this tree (null -> the process-wide default context)
inline Context context_of(CUstream_st* stream) {
  if (!stream) return StreamTempBufferRoundRobin::default_instance();
  return reinterpret_cast<StreamTempBufferRoundRobin*>(stream)->shared_from_this();
}
}  // namespace glim_amd

namespace gtsam_points {

class CUDAStream {
public:
  CUDAStream() : ctx_(std::make_shared<glim_amd::StreamTempBufferRoundRobin>(1, 0)) {}
  ~CUDAStream() {}
  CUDAStream(const CUDAStream&) = delete;
  CUDAStream& operator=(const CUDAStream&) = delete;
  operator CUstream_st*() { return reinterpret_cast<CUstream_st*>(ctx_.get()); }
  operator CUstream_st*() const { return reinterpret_cast<CUstream_st*>(ctx_.get()); }
  void sync() { glim_amd_ctx_synchronize(ctx_->context()); }
  const glim_amd::Context& context() const { return ctx_; }

private:
  glim_amd::Context ctx_;
};

}  // namespace gtsam_points
