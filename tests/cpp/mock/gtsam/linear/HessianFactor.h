#pragma once
#include <gtsam/base/Matrix.h>
#include <gtsam/inference/Key.h>
#include <memory>
namespace gtsam {
class GaussianFactor {
public:
  using shared_ptr = std::shared_ptr<GaussianFactor>;
  virtual ~GaussianFactor() {}
};
// error(x) = 0.5 (f - 2 x^T g + x^T G x)
class HessianFactor : public GaussianFactor {
public:
  HessianFactor(Key j, const Matrix& G, const Vector& g, double f) : keys{j}, G11(G), g1(g), f(f) {}
  HessianFactor(Key j1, Key j2, const Matrix& G11, const Matrix& G12, const Vector& g1, const Matrix& G22, const Vector& g2, double f)
  : keys{j1, j2}, G11(G11), G12(G12), G22(G22), g1(g1), g2(g2), f(f) {}
  KeyVector keys;
  Matrix G11, G12, G22;
  Vector g1, g2;
  double f;
};
}  // namespace gtsam
