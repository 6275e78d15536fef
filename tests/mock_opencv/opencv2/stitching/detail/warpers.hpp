// DECLARATIONS ONLY - see tests/mock_opencv/README.md
#pragma once
#include "../../core.hpp"
namespace cv { namespace detail {
class RotationWarper {
  public:
    virtual ~RotationWarper();
    virtual Rect buildMaps(Size src_size, InputArray K, InputArray R, OutputArray xmap, OutputArray ymap) = 0;
    virtual Point warp(InputArray src, InputArray K, InputArray R, int interp_mode, int border_mode, OutputArray dst) = 0;
    virtual Rect warpRoi(Size src_size, InputArray K, InputArray R) = 0;
};
class SphericalWarper : public RotationWarper {
  public:
    SphericalWarper(float scale);
    Rect buildMaps(Size, InputArray, InputArray, OutputArray, OutputArray); Point warp(InputArray, InputArray, InputArray, int, int, OutputArray); Rect warpRoi(Size, InputArray, InputArray);
};
class CylindricalWarper : public RotationWarper {
  public:
    CylindricalWarper(float scale);
    Rect buildMaps(Size, InputArray, InputArray, OutputArray, OutputArray); Point warp(InputArray, InputArray, InputArray, int, int, OutputArray); Rect warpRoi(Size, InputArray, InputArray);
};
}}
