// glim_io_compat.hpp -- header-only C++17 readers / writers of the on-disk map formats GLIM's offline tools exchange
// (SURVEY.md 8f rank 4), so that maps built through this library round-trip through `offline_viewer`:
//
//   <dump>/graph.txt            GlobalMapping::save  src/glim/mapping/global_mapping.cpp:576-598, ::load :690-711
//   <dump>/<%06d>/data.txt      SubMap::save         src/glim/mapping/sub_map.cpp:24-62,          ::load :76-141
//   <dump>/<%06d>/*_compact.bin gtsam_points::PointCloud::save_compact (sub_map.cpp:62) -> glim_amd_cloud_save_compact (C ABI)
//
// graph.bin / values.bin are GTSAM boost-serialisation archives of the non-matching-cost factors and the estimate; they are
// written by GTSAM itself and are outside this library (it never sees those factors).
// Matrices are printed the way `ofs << Eigen::Matrix` prints them (Eigen's default IOFormat: stream precision, coefficients
// right-aligned to the widest one, single-space separator), so the files are byte-compatible, not merely parseable.
#pragma once

#include <array>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "gtsam_points_compat.hpp"

namespace glim_amd {

// `os << Eigen::Matrix<double, rows, cols>` (row-major input here)
inline void write_eigen_matrix(std::ostream& os, const double* m, int rows, int cols) {
  std::vector<std::string> cell((std::size_t)rows * cols);
  std::size_t width = 0;
  for (int i = 0; i < rows * cols; i++) {
    std::ostringstream ss;
    ss.copyfmt(os);  // Eigen formats every coefficient with the stream's own precision / flags
    ss.width(0);
    ss << m[i];
    cell[(std::size_t)i] = ss.str();
    width = std::max(width, cell[(std::size_t)i].size());
  }
  for (int r = 0; r < rows; r++) {
    if (r) os << "\n";
    for (int c = 0; c < cols; c++) {
      if (c) os << " ";
      os << std::string(width - cell[(std::size_t)(r * cols + c)].size(), ' ') << cell[(std::size_t)(r * cols + c)];
    }
  }
}

inline void write_isometry(std::ostream& os, const Isometry3d& T) {  // `os << T.matrix()` (4x4)
  double m[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
  for (int i = 0; i < 12; i++) m[i] = T.m[(std::size_t)i];
  write_eigen_matrix(os, m, 4, 4);
}
inline bool read_isometry(std::istream& is, Isometry3d& T) {
  double m[16];
  for (double& v : m)
    if (!(is >> v)) return false;
  for (int i = 0; i < 12; i++) T.m[(std::size_t)i] = m[i];
  return true;
}

// SubMap::save / ::load, text part (data.txt)
struct SubMapData {
  struct Frame {
    long id = 0;
    double stamp = 0.0;
    Isometry3d T_odom_lidar, T_world_lidar;
    std::array<double, 3> v_world_imu{{0, 0, 0}};
  };
  int id = 0;
  Isometry3d T_world_origin, T_origin_endpoint_L, T_origin_endpoint_R, T_lidar_imu;
  std::array<double, 6> imu_bias{{0, 0, 0, 0, 0, 0}};
  int frame_id = 0;  // FrameID of the last frame (sub_map.cpp:34)
  std::vector<Frame> frames;

  bool save(const std::string& dir) const {  // sub_map.cpp:24-49
    std::ofstream ofs(dir + "/data.txt");
    if (!ofs) return false;
    ofs << "id: " << id << std::endl;
    ofs << "T_world_origin: " << std::endl;
    write_isometry(ofs, T_world_origin);
    ofs << std::endl << "T_origin_endpoint_L: " << std::endl;
    write_isometry(ofs, T_origin_endpoint_L);
    ofs << std::endl << "T_origin_endpoint_R: " << std::endl;
    write_isometry(ofs, T_origin_endpoint_R);
    ofs << std::endl;
    if (!frames.empty()) {
      ofs << "T_lidar_imu: " << std::endl;
      write_isometry(ofs, T_lidar_imu);
      ofs << std::endl << "imu_bias: ";
      write_eigen_matrix(ofs, imu_bias.data(), 1, 6);
      ofs << std::endl << "frame_id: " << frame_id << std::endl;
    }
    ofs << "num_frames: " << frames.size() << std::endl;
    for (std::size_t i = 0; i < frames.size(); i++) {
      char stamp[64];
      std::snprintf(stamp, sizeof(stamp), "%.9f", frames[i].stamp);  // boost::format("%.9f")
      ofs << "frame_" << i << std::endl;
      ofs << "id: " << frames[i].id << std::endl;
      ofs << "stamp: " << stamp << std::endl;
      ofs << "T_odom_lidar: " << std::endl;
      write_isometry(ofs, frames[i].T_odom_lidar);
      ofs << std::endl << "T_world_lidar: " << std::endl;
      write_isometry(ofs, frames[i].T_world_lidar);
      ofs << std::endl << "v_world_imu: ";
      write_eigen_matrix(ofs, frames[i].v_world_imu.data(), 1, 3);
      ofs << std::endl;
    }
    return (bool)ofs;
  }

  bool load(const std::string& dir) {  // sub_map.cpp:76-141 (same token-skipping reads)
    std::ifstream ifs(dir + "/data.txt");
    if (!ifs) return false;
    std::string token;
    ifs >> token >> id;
    ifs >> token;
    if (!read_isometry(ifs, T_world_origin)) return false;
    ifs >> token;
    if (!read_isometry(ifs, T_origin_endpoint_L)) return false;
    ifs >> token;
    if (!read_isometry(ifs, T_origin_endpoint_R)) return false;
    ifs >> token;
    if (!read_isometry(ifs, T_lidar_imu)) return false;
    ifs >> token;
    for (double& v : imu_bias) ifs >> v;
    ifs >> token >> frame_id;
    int num_frames = 0;
    ifs >> token >> num_frames;
    if (!ifs || num_frames < 0) return false;
    frames.assign((std::size_t)num_frames, Frame());
    for (auto& f : frames) {
      ifs >> token >> token >> f.id;
      ifs >> token >> f.stamp;
      ifs >> token;
      if (!read_isometry(ifs, f.T_odom_lidar)) return false;
      ifs >> token;
      if (!read_isometry(ifs, f.T_world_lidar)) return false;
      ifs >> token;
      for (double& v : f.v_world_imu) ifs >> v;
    }
    return (bool)ifs;
  }
};

// graph.txt: the matching-cost factors GLIM cannot serialise through GTSAM and re-creates on load
struct GraphTxt {
  int num_submaps = 0;
  int num_all_frames = 0;
  std::vector<std::tuple<std::string, int, int>> matching_cost_factors;  // (type, first submap, second submap); type: gicp / vgicp / vgicp_gpu

  bool save(const std::string& dir) const {  // global_mapping.cpp:576-598
    std::ofstream ofs(dir + "/graph.txt");
    if (!ofs) return false;
    ofs << "num_submaps: " << num_submaps << std::endl;
    ofs << "num_all_frames: " << num_all_frames << std::endl;
    ofs << "num_matching_cost_factors: " << matching_cost_factors.size() << std::endl;
    for (const auto& f : matching_cost_factors) ofs << "matching_cost " << std::get<0>(f) << " " << std::get<1>(f) << " " << std::get<2>(f) << std::endl;
    return (bool)ofs;
  }
  bool load(const std::string& dir) {  // global_mapping.cpp:690-711
    std::ifstream ifs(dir + "/graph.txt");
    if (!ifs) return false;
    std::string token;
    int n = 0;
    ifs >> token >> num_submaps;
    ifs >> token >> num_all_frames;
    ifs >> token >> n;
    if (!ifs || n < 0) return false;
    matching_cost_factors.resize((std::size_t)n);
    for (auto& f : matching_cost_factors) ifs >> token >> std::get<0>(f) >> std::get<1>(f) >> std::get<2>(f);
    return (bool)ifs;
  }
};

// "<dump>/%06d" (global_mapping.cpp:631, :716)
inline std::string submap_dir(const std::string& dump, int index) {
  char buf[16];
  std::snprintf(buf, sizeof(buf), "%06d", index);
  return dump + "/" + buf;
}

// frame->save_compact(path) / PointCloudCPU::load(path) of a submap's merged cloud, device side
inline void save_compact(const PointCloudGPU& cloud, const std::string& dir) { check(glim_amd_cloud_save_compact(cloud.handle(), dir.c_str()), "save_compact"); }
inline PointCloudGPU::Ptr load_compact(const std::string& dir, Context ctx = nullptr) {
  ctx = ctx ? ctx : StreamTempBufferRoundRobin::default_instance();
  glim_amd_cloud* h = nullptr;
  check(glim_amd_cloud_load_compact(ctx->context(), dir.c_str(), &h), "load_compact");
  return PointCloudGPU::adopt(h, ctx);
}

}  // namespace glim_amd
