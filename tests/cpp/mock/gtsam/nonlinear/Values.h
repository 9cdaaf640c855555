#pragma once
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Key.h>
#include <map>
#include <stdexcept>
namespace gtsam {
class Values {  // only Pose3 values are needed here
public:
  void insert(Key k, const Pose3& p) { v_[k] = p; }
  bool exists(Key k) const { return v_.count(k) != 0; }
  template <class T>
  const T& at(Key k) const {
    auto it = v_.find(k);
    if (it == v_.end()) throw std::out_of_range("ValuesKeyDoesNotExist");
    return it->second;
  }

private:
  std::map<Key, Pose3> v_;
};
}  // namespace gtsam
