#pragma once
#include <memory>
#include <gtsam/base/Matrix.h>
namespace gtsam {
namespace noiseModel {
class Base {
public:
  typedef std::shared_ptr<Base> shared_ptr;
  virtual ~Base() {}
};
class Gaussian : public Base {
public:
  typedef std::shared_ptr<Gaussian> shared_ptr;
  static shared_ptr Information(const Matrix&);
  static shared_ptr Covariance(const Matrix&);
};
class Diagonal : public Gaussian {
public:
  typedef std::shared_ptr<Diagonal> shared_ptr;
};
class Isotropic : public Diagonal {
public:
  typedef std::shared_ptr<Isotropic> shared_ptr;
  static shared_ptr Sigma(size_t dim, double sigma);
  static shared_ptr Precision(size_t dim, double precision);
};
namespace mEstimator {
class Base {
public:
  typedef std::shared_ptr<Base> shared_ptr;
  virtual ~Base() {}
};
class Huber : public Base {
public:
  static shared_ptr Create(double k);
};
class Cauchy : public Base {
public:
  static shared_ptr Create(double k);
};
class Tukey : public Base {
public:
  static shared_ptr Create(double k);
};
}  // namespace mEstimator
class Robust : public Base {
public:
  typedef std::shared_ptr<Robust> shared_ptr;
  static shared_ptr Create(const mEstimator::Base::shared_ptr&, const Base::shared_ptr&);
};
}  // namespace noiseModel
typedef noiseModel::Base::shared_ptr SharedNoiseModel;
}  // namespace gtsam
