#pragma once
#include <memory>
#include <gtsam/navigation/ImuBias.h>
#include <gtsam/navigation/NavState.h>
#include <gtsam/nonlinear/NonlinearFactor.h>
namespace gtsam {
class PreintegrationParams {
public:
  static std::shared_ptr<PreintegrationParams> MakeSharedU(double g = 9.81);
  static std::shared_ptr<PreintegrationParams> MakeSharedD(double g = 9.81);
};
class PreintegratedImuMeasurements {
public:
  PreintegratedImuMeasurements();
  NavState predict(const NavState&, const imuBias::ConstantBias&) const;
  void resetIntegrationAndSetBias(const imuBias::ConstantBias&);
  void integrateMeasurement(const Vector3&, const Vector3&, double);
};
class ImuFactor : public NonlinearFactor {
public:
  ImuFactor(Key, Key, Key, Key, Key, const PreintegratedImuMeasurements&);
  size_t dim() const override;
  double error(const Values&) const override;
  std::shared_ptr<GaussianFactor> linearize(const Values&) const override;
  shared_ptr clone() const override;
};
}  // namespace gtsam
