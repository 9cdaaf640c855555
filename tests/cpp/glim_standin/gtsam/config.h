#pragma once
// stand-in of GTSAM's generated config.h (what src/glim/util/debug.cpp prints)
#define GTSAM_VERSION_MAJOR 4
#define GTSAM_VERSION_MINOR 3
#define GTSAM_VERSION_PATCH 0
#define GTSAM_VERSION_STRING "4.3a0"
