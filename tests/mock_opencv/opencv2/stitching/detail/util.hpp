// DECLARATIONS ONLY - see tests/mock_opencv/README.md
#pragma once
#include "../../core.hpp"
namespace cv { namespace detail { Rect resultRoi(const std::vector<Point>& corners, const std::vector<Size>& sizes); }}
