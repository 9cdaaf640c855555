// stand-in for spdlog (see ../Eigen/Core): the reference only logs a critical message before abort() on these paths
#pragma once
#include <cstdio>
#include <cstdlib>
namespace spdlog {
template <class... A>
inline void critical(const char* msg, A&&...) { std::fprintf(stderr, "[critical] %s\n", msg); }
template <class... A>
inline void warn(const char* msg, A&&...) { std::fprintf(stderr, "[warn] %s\n", msg); }
template <class... A>
inline void trace(const char*, A&&...) {}
template <class... A>
inline void debug(const char*, A&&...) {}
}  // namespace spdlog
