#pragma once
#include <fstream>
#include <iostream>
#include <string>
#include <exception>
#include <ostream>
#include <set>
namespace boost {
class format {  // boost/format.hpp reaches GLIM's sources through other boost / GTSAM headers
public:
  explicit format(const char*);
  template <class T>
  format& operator%(const T&);
  std::string str() const;
};
std::ostream& operator<<(std::ostream&, const format&);
namespace archive {
class archive_exception : public std::exception {};
}  // namespace archive
namespace filesystem {
class path {
public:
  path() {}
  path(const std::string& s) : s_(s) {}
  path(const char* s) : s_(s) {}
  std::string string() const { return s_; }

private:
  std::string s_;
};
bool exists(const path&);
bool create_directories(const path&);
bool create_directory(const path&);
}  // namespace filesystem
}  // namespace boost
