// stand-in for fmt: only reached on the reference's error paths (exception messages)
#pragma once
#include <sstream>
#include <string>
namespace fmt {
template <class... A>
inline std::string format(const char* f, A&&...) { return std::string(f); }
}  // namespace fmt
