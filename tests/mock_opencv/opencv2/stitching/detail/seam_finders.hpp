// DECLARATIONS ONLY - see tests/mock_opencv/README.md
#pragma once
#include "../../core.hpp"
namespace cv { namespace detail {
class SeamFinder { public: virtual ~SeamFinder(); virtual void find(const std::vector<UMat>& src, const std::vector<Point>& corners, std::vector<UMat>& masks) = 0; };
class VoronoiSeamFinder : public SeamFinder { public: void find(const std::vector<UMat>& src, const std::vector<Point>& corners, std::vector<UMat>& masks); };
class GraphCutSeamFinderBase { public: enum CostType { COST_COLOR, COST_COLOR_GRAD }; };
class GraphCutSeamFinder : public GraphCutSeamFinderBase, public SeamFinder {
  public:
    GraphCutSeamFinder(int cost_type = COST_COLOR_GRAD, float terminal_cost = 10000.f, float bad_region_penalty = 1000.f);
    void find(const std::vector<UMat>& src, const std::vector<Point>& corners, std::vector<UMat>& masks);
};
}}
