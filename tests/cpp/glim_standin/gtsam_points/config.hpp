#pragma once
// stand-in of the generated gtsam_points/config.hpp: a CPU-only gtsam_points install leaves GTSAM_POINTS_USE_CUDA undefined; a HIP build of
// libglim defines it on the command line (adapters/gtsam_points_hip/README.md)
#define GTSAM_POINTS_VERSION_MAJOR 1
#define GTSAM_POINTS_VERSION_MINOR 2
#define GTSAM_POINTS_VERSION_PATCH 2
#define GTSAM_POINTS_VERSION_STRING "1.2.2"
#define GTSAM_POINTS_GIT_HASH "stand-in"
