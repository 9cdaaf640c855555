#pragma once
#include <Eigen/Core>
namespace gtsam {
using Matrix = Eigen::MatrixXd;
using Vector = Eigen::VectorXd;
}  // namespace gtsam
