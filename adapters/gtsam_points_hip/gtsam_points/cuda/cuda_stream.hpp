// gtsam_points/cuda/cuda_stream.hpp, MI355X edition: gtsam_points::CUDAStream as GLIM uses it -- a RAII stream handed to voxel maps and
// overlap_gpu (src/glim/odometry/odometry_estimation_gpu.cpp:76, :103, :231; sub_mapping.cpp:86; global_mapping.cpp:109).
// HIP streams belong to the library's context pool (glim_amd_ctx), so this object only carries an opaque tag: every call that receives it
// runs on the streams of the default context.
#pragma once

struct CUstream_st;  // the reference's stream handle type; never dereferenced here

namespace gtsam_points {

class CUDAStream {
public:
  CUDAStream() {}
  ~CUDAStream() {}
  CUDAStream(const CUDAStream&) = delete;
  CUDAStream& operator=(const CUDAStream&) = delete;
  operator CUstream_st*() { return nullptr; }
  operator CUstream_st*() const { return nullptr; }
  void sync() {}
};

}  // namespace gtsam_points
