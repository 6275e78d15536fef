// DECLARATIONS ONLY - see tests/mock_opencv/README.md
#pragma once
#include "core.hpp"
namespace cv {
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_LINEAR_EXACT = 5 };
void resize(InputArray, OutputArray, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void dilate(InputArray, OutputArray, InputArray kernel);
void pyrDown(InputArray, OutputArray, const Size& dstsize = Size());
void pyrUp(InputArray, OutputArray, const Size& dstsize = Size());
void rectangle(InputOutputArray, Rect, const Scalar&, int thickness = 1);
}
