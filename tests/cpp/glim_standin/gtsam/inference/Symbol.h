#pragma once
#include <gtsam/inference/Key.h>
namespace gtsam {
namespace symbol_shorthand {
inline Key B(std::uint64_t j) { return (std::uint64_t('b') << 56) | j; }
inline Key V(std::uint64_t j) { return (std::uint64_t('v') << 56) | j; }
inline Key X(std::uint64_t j) { return (std::uint64_t('x') << 56) | j; }
}  // namespace symbol_shorthand
}  // namespace gtsam
