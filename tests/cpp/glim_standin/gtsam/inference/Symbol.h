#pragma once
#include <string>
#include <gtsam/inference/Key.h>
namespace gtsam {
class Symbol {
public:
  Symbol(unsigned char c, std::uint64_t j) : c_(c), j_(j) {}
  Symbol(Key key) : c_((unsigned char)(key >> 56)), j_(key & 0x00ffffffffffffffull) {}
  operator Key() const { return (std::uint64_t(c_) << 56) | j_; }
  Key key() const { return (std::uint64_t(c_) << 56) | j_; }
  unsigned char chr() const { return c_; }
  std::uint64_t index() const { return j_; }
  operator std::string() const;
  std::string string() const;

private:
  unsigned char c_;
  std::uint64_t j_;
};
namespace symbol_shorthand {
inline Key B(std::uint64_t j) { return (std::uint64_t('b') << 56) | j; }
inline Key E(std::uint64_t j) { return (std::uint64_t('e') << 56) | j; }
inline Key V(std::uint64_t j) { return (std::uint64_t('v') << 56) | j; }
inline Key X(std::uint64_t j) { return (std::uint64_t('x') << 56) | j; }
}  // namespace symbol_shorthand
}  // namespace gtsam
