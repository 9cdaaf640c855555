#pragma once
#include <gtsam/slam/BetweenFactor.h>
