#pragma once
namespace cv {
class Mat {};
}  // namespace cv
