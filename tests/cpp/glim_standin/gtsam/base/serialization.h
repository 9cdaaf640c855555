#pragma once
#include <string>
namespace gtsam {
template <class T>
bool serializeToBinaryFile(const T&, const std::string&);
template <class T>
bool deserializeFromBinaryFile(const std::string&, T&);
}  // namespace gtsam
