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
  // error(x) = 0.5 (f - 2 x^T g + x^T G x).  The blocks are really stored: tests/cpp/test_shim_cpu_names.cpp runs a Gauss-Newton loop on them.
  HessianFactor(Key j, const Matrix& G, const Vector& g, double f) : keys_{j}, G22_(G), g2_(g), f_(f) {}
  HessianFactor(Key j1, Key j2, const Matrix& G11, const Matrix& G12, const Vector& g1, const Matrix& G22, const Vector& g2, double f)
  : keys_{j1, j2}, G11_(G11), G12_(G12), G22_(G22), g1_(g1), g2_(g2), f_(f) {}
  KeyVector keys_;
  Matrix G11_, G12_, G22_;  // (unary: G22_ / g2_ are the block of the only key)
  Vector g1_, g2_;
  double f_ = 0.0;
};
class GaussianFactorGraph {
public:
  std::map<Key, Matrix> hessianBlockDiagonal() const;
};
}  // namespace gtsam
