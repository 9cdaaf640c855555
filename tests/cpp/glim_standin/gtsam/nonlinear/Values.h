#pragma once
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Key.h>
#include <map>
namespace gtsam {
class Value {};
class Values {
public:
  struct KeyValuePair {
    Key key;
    const Value& value;
  };
  struct const_iterator {
    const KeyValuePair* operator->() const;
    const KeyValuePair& operator*() const;
    const_iterator& operator++();
    bool operator!=(const const_iterator&) const;
    bool operator==(const const_iterator&) const;
  };
  const_iterator begin() const;
  const_iterator end() const;
  void insert(Key k, const Value& v);
  void insert_or_assign(const Values&);
  Values() {}
  bool exists(Key k) const { return poses_.count(k) != 0; }
  template <class T>
  const T at(Key k) const;
  template <class T>
  void insert(Key k, const T& v);
  void insert(const Values&);
  template <class T>
  void update(Key k, const T& v);
  template <class T>
  void insert_or_assign(Key k, const T& v);
  void erase(Key k);
  void clear();
  bool empty() const;
  size_t size() const;
  KeyVector keys() const;

private:
  std::map<Key, Pose3> poses_;  // the poses are really stored: tests/cpp/test_shim.cpp runs against these stand-ins
};
template <>
inline const Pose3 Values::at<Pose3>(Key k) const { return poses_.at(k); }
template <>
inline void Values::insert<Pose3>(Key k, const Pose3& v) { poses_[k] = v; }
}  // namespace gtsam
