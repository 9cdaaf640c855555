// test_io.cpp -- the on-disk text formats of include/glim_amd/glim_io_compat.hpp (graph.txt, data.txt) on the CPU: known-answer
// bytes for Eigen's matrix printing, a write -> read round trip, and a parse of a data.txt laid out like SubMap::save writes it.
// Built and run by tests/test_io.py:  g++ -std=c++17 test_io.cpp -lglim_amd
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "../../include/glim_amd/glim_io_compat.hpp"

using namespace glim_amd;

#define REQUIRE(cond)                                                   \
  do {                                                                  \
    if (!(cond)) {                                                      \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

static std::string slurp(const std::string& path) {
  std::ifstream f(path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

int main(int argc, char** argv) {
  REQUIRE(argc == 2);
  const std::string dir = argv[1];

  // Eigen's default IOFormat: stream precision (6 significant digits), every coefficient right-aligned to the widest one
  {
    std::ostringstream os;
    const double m[4] = {1.0, -0.5, 10.25, 3.0};
    write_eigen_matrix(os, m, 2, 2);
    REQUIRE(os.str() == "    1  -0.5\n10.25     3");
    std::ostringstream id;
    write_isometry(id, Isometry3d::Identity());
    REQUIRE(id.str() == "1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1");
    std::ostringstream row;
    const double v[3] = {0.123456789, -2.0, 1e-7};
    write_eigen_matrix(row, v, 1, 3);
    REQUIRE(row.str() == "0.123457       -2    1e-07");
  }

  // data.txt round trip (6 significant digits survive, like the reference's own save -> load)
  SubMapData sm;
  sm.id = 7;
  sm.T_world_origin.m = {0.5, -0.866025, 0, 12.5, 0.866025, 0.5, 0, -3.25, 0, 0, 1, 0.75};
  sm.T_origin_endpoint_L.m = {1, 0, 0, -1.5, 0, 1, 0, 0.25, 0, 0, 1, 0};
  sm.T_origin_endpoint_R.m = {1, 0, 0, 1.5, 0, 1, 0, -0.25, 0, 0, 1, 0};
  sm.T_lidar_imu.m = {1, 0, 0, 0.006, 0, 1, 0, -0.012, 0, 0, 1, 0.008};
  sm.imu_bias = {0.001, -0.002, 0.003, 1e-5, -2e-5, 3e-5};
  sm.frame_id = 2;
  for (int i = 0; i < 3; i++) {
    SubMapData::Frame f;
    f.id = 100 + i;
    f.stamp = 1700000000.123456789 + 0.1 * i;
    f.T_odom_lidar.m = {1, 0, 0, 0.1 * i, 0, 1, 0, 0, 0, 0, 1, 0};
    f.T_world_lidar.m = {1, 0, 0, 0.1 * i + 5, 0, 1, 0, 1, 0, 0, 1, 0};
    f.v_world_imu = {1.5, -0.25, 0.0};
    sm.frames.push_back(f);
  }
  REQUIRE(sm.save(dir));
  const std::string text = slurp(dir + "/data.txt");
  REQUIRE(text.rfind("id: 7\nT_world_origin: \n", 0) == 0);
  REQUIRE(text.find("\nimu_bias:  0.001 -0.002  0.003  1e-05 -2e-05  3e-05\nframe_id: 2\nnum_frames: 3\nframe_0\nid: 100\nstamp: 1700000000.123456717\n") != std::string::npos);
  REQUIRE(text.find("\nv_world_imu:   1.5 -0.25     0\n") != std::string::npos);
  SubMapData back;
  REQUIRE(back.load(dir));
  REQUIRE(back.id == 7 && back.frame_id == 2 && back.frames.size() == 3);
  for (int i = 0; i < 12; i++) {
    REQUIRE(std::fabs(back.T_world_origin.m[i] - sm.T_world_origin.m[i]) <= 1e-5 * (1.0 + std::fabs(sm.T_world_origin.m[i])));
    REQUIRE(back.T_lidar_imu.m[i] == sm.T_lidar_imu.m[i]);
  }
  for (int i = 0; i < 6; i++) REQUIRE(back.imu_bias[i] == sm.imu_bias[i]);
  for (int i = 0; i < 3; i++) {
    REQUIRE(back.frames[i].id == 100 + i);
    REQUIRE(std::fabs(back.frames[i].stamp - sm.frames[i].stamp) < 1e-6);
    REQUIRE(back.frames[i].T_world_lidar.m[3] == sm.frames[i].T_world_lidar.m[3]);
    REQUIRE(back.frames[i].v_world_imu[1] == -0.25);
  }

  // graph.txt round trip + exact bytes
  GraphTxt g;
  g.num_submaps = 3;
  g.num_all_frames = 45;
  g.matching_cost_factors = {{"vgicp_gpu", 0, 1}, {"vgicp_gpu", 0, 2}, {"gicp", 1, 2}};
  REQUIRE(g.save(dir));
  REQUIRE(slurp(dir + "/graph.txt") ==
          "num_submaps: 3\nnum_all_frames: 45\nnum_matching_cost_factors: 3\nmatching_cost vgicp_gpu 0 1\nmatching_cost vgicp_gpu 0 2\nmatching_cost gicp 1 2\n");
  GraphTxt h;
  REQUIRE(h.load(dir));
  REQUIRE(h.num_submaps == 3 && h.num_all_frames == 45 && h.matching_cost_factors.size() == 3);
  REQUIRE(std::get<0>(h.matching_cost_factors[2]) == "gicp" && std::get<1>(h.matching_cost_factors[2]) == 1 && std::get<2>(h.matching_cost_factors[2]) == 2);
  REQUIRE(submap_dir("/tmp/dump", 12) == "/tmp/dump/000012");
  REQUIRE(!GraphTxt().load(dir + "/missing"));
  // row a9 host helpers: median range of <= max_scan_count strided samples and the resolution blend
  {
    std::vector<double> pts;
    for (int i = 0; i < 1000; i++) {
      const double r = 1.0 + 0.01 * i;
      pts.insert(pts.end(), {r, 0.0, 0.0, 1.0});
    }
    REQUIRE(std::fabs(median_distance(pts.data(), 1000, 256) - (1.0 + 0.01 * 501)) < 1e-12);  // step 3 -> 334 samples, element 167 -> i = 501
    REQUIRE(std::fabs(median_distance(pts.data(), 100, 256) - (1.0 + 0.01 * 50)) < 1e-12);
    REQUIRE(median_distance(pts.data(), 0) == 0.0);
    REQUIRE(std::fabs(adaptive_voxel_resolution(8.0, 0.25, 0.5, 4.0, 12.0) - 0.375) < 1e-15);
    REQUIRE(adaptive_voxel_resolution(1.0, 0.25, 0.5, 4.0, 12.0) == 0.25 && adaptive_voxel_resolution(99.0, 0.25, 0.5, 4.0, 12.0) == 0.5);
  }
  std::printf("test_io OK\n");
  return 0;
}
