#pragma once
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Key.h>
#include <map>
namespace gtsam {
class Values {
public:
  bool exists(Key k) const { return poses_.count(k) != 0; }
  template <class T>
  const T& at(Key k) const { return poses_.at(k); }
  void insert(Key k, const Pose3& p) { poses_[k] = p; }

private:
  std::map<Key, Pose3> poses_;
};
}  // namespace gtsam
