#pragma once
#include <cstdint>
#include <vector>
namespace gtsam {
using Key = std::uint64_t;
using KeyVector = std::vector<Key>;
}  // namespace gtsam
