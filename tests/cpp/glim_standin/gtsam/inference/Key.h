#pragma once
#include <cstdint>
#include <vector>
namespace gtsam {
typedef std::uint64_t Key;
typedef std::vector<Key> KeyVector;
}  // namespace gtsam
