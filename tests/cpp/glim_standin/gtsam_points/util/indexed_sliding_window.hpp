#pragma once
#include <deque>
namespace gtsam_points {
template <class T>
class IndexedSlidingWindow {
public:
  T& operator[](int i) { return data_[(size_t)i]; }
  const T& operator[](int i) const { return data_[(size_t)i]; }
  void clear() { data_.clear(); }
  int size() const { return (int)data_.size(); }
  T& back() { return data_.back(); }
  const T& back() const { return data_.back(); }

private:
  std::deque<T> data_;
};
}  // namespace gtsam_points
