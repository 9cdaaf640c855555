#pragma once
#include <gtsam/base/Matrix.h>
#include <gtsam/inference/Key.h>
#include <map>
#include <memory>
namespace gtsam {
class GaussianFactor {
public:
  typedef std::shared_ptr<GaussianFactor> shared_ptr;
  virtual ~GaussianFactor() {}
  std::map<Key, Matrix> hessianBlockDiagonal() const;
};
class HessianFactor : public GaussianFactor {
public:
  HessianFactor(Key, const Matrix&, const Vector&, double) {}
  HessianFactor(Key, Key, const Matrix&, const Matrix&, const Vector&, const Matrix&, const Vector&, double) {}
};
class GaussianFactorGraph {
public:
  std::map<Key, Matrix> hessianBlockDiagonal() const;
};
}  // namespace gtsam
