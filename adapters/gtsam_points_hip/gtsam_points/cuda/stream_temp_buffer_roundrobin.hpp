// gtsam_points/cuda/stream_temp_buffer_roundrobin.hpp, MI355X edition (odometry_estimation_gpu.cpp:77,139; sub_mapping.cpp:87,296-299;
// global_mapping.cpp:110,331,462,855): GLIM takes a (stream, scratch buffer) pair per factor from this pool.  Here the pool OWNS a library
// context with `num_streams` HIP streams; the stream handle of every pair is that context (factor sets take their streams from it round
// robin: glim_amd_ctx_create), the scratch buffer is a tag -- scratch memory belongs to the library's device pool.
#pragma once

#include <memory>
#include <utility>

#include <gtsam_points/cuda/cuda_stream.hpp>

namespace gtsam_points {

class TempBufferManager {};

class StreamTempBufferRoundRobin {
public:
  explicit StreamTempBufferRoundRobin(int num_streams = 8)
  : num_streams_(num_streams), ctx_(std::make_shared<glim_amd::StreamTempBufferRoundRobin>(num_streams, 0)), buffer_(std::make_shared<TempBufferManager>()) {}
  std::pair<CUstream_st*, std::shared_ptr<TempBufferManager>> get_stream_buffer() { return {reinterpret_cast<CUstream_st*>(ctx_.get()), buffer_}; }
  int num_streams() const { return num_streams_; }
  const glim_amd::Context& context() const { return ctx_; }

private:
  int num_streams_;
  glim_amd::Context ctx_;
  std::shared_ptr<TempBufferManager> buffer_;
};

}  // namespace gtsam_points
