// gtsam_points/cuda/stream_temp_buffer_roundrobin.hpp, MI355X edition (odometry_estimation_gpu.cpp:77,139; sub_mapping.cpp:87,296-299;
// global_mapping.cpp:110,331,462,855): GLIM takes a (stream, scratch buffer) pair per factor from this pool.  In this library streams and
// scratch live inside the context (glim_amd_ctx_create(device, num_streams, ...)): a factor SET picks its stream from that pool, so the pair
// handed out here is a placeholder the factor constructors accept and ignore.
#pragma once

#include <memory>
#include <utility>

#include <gtsam_points/cuda/cuda_stream.hpp>

namespace gtsam_points {

class TempBufferManager {};

class StreamTempBufferRoundRobin {
public:
  explicit StreamTempBufferRoundRobin(int num_streams = 8) : num_streams_(num_streams), buffer_(std::make_shared<TempBufferManager>()) {}
  std::pair<CUstream_st*, std::shared_ptr<TempBufferManager>> get_stream_buffer() { return {nullptr, buffer_}; }
  int num_streams() const { return num_streams_; }

private:
  int num_streams_;
  std::shared_ptr<TempBufferManager> buffer_;
};

}  // namespace gtsam_points
