#pragma once
#include <gtsam_points/types/gaussian_voxelmap.hpp>
