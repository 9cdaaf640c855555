#pragma once
#include <sstream>
#include <string>
namespace fmt {
template <class... A>
std::string format(const char*, A&&...);  // declared only: the compile tests never link
}
