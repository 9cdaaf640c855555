#pragma once
#include <gtsam/nonlinear/NonlinearFactor.h>
namespace gtsam {
namespace imuBias {
class ConstantBias {};
}  // namespace imuBias
class NavState {};
class PreintegratedImuMeasurements {};
class ImuFactor {};
}  // namespace gtsam
