#pragma once
#include <gtsam/nonlinear/NonlinearFactor.h>
