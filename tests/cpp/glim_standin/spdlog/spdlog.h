#pragma once
#include <cstdio>
#include <cstdlib>
namespace spdlog {
template <class... A> inline void critical(const char*, A&&...) {}
template <class... A> inline void error(const char*, A&&...) {}
template <class... A> inline void warn(const char*, A&&...) {}
template <class... A> inline void info(const char*, A&&...) {}
template <class... A> inline void debug(const char*, A&&...) {}
}  // namespace spdlog
