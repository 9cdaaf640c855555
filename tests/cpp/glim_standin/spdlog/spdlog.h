#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
namespace spdlog {
template <class... A> inline void critical(const char*, A&&...) {}
template <class... A> inline void error(const char*, A&&...) {}
template <class... A> inline void warn(const char*, A&&...) {}
template <class... A> inline void info(const char*, A&&...) {}
template <class... A> inline void debug(const char*, A&&...) {}
template <class... A> inline void trace(const char*, A&&...) {}
class logger {
public:
  void critical(const std::string&) {}
  void error(const std::string&) {}
  void warn(const std::string&) {}
  void info(const std::string&) {}
  void debug(const std::string&) {}
  void trace(const std::string&) {}
  template <class... A> void critical(const char*, A&&...) {}
  template <class... A> void error(const char*, A&&...) {}
  template <class... A> void warn(const char*, A&&...) {}
  template <class... A> void info(const char*, A&&...) {}
  template <class... A> void debug(const char*, A&&...) {}
  template <class... A> void trace(const char*, A&&...) {}
};
namespace sinks {
class ringbuffer_sink_mt;
}
}  // namespace spdlog
