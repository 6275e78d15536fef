// DECLARATIONS ONLY - see tests/mock_opencv/README.md
#pragma once
#include "core.hpp"
namespace cv { enum { IMREAD_COLOR = 1 }; Mat imread(const String&, int flags = IMREAD_COLOR); bool imwrite(const String&, InputArray); }
