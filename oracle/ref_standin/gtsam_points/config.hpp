// stand-in for gtsam_points/config.hpp: OpenMP build, no TBB (GTSAM_POINTS_USE_TBB undefined)
#pragma once
#define GTSAM_POINTS_USE_OPENMP
