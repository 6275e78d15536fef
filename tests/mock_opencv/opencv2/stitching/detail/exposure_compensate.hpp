// DECLARATIONS ONLY - see tests/mock_opencv/README.md
#pragma once
#include "../../core.hpp"
namespace cv { namespace detail {
class ExposureCompensator {
  public:
    virtual ~ExposureCompensator();
    enum { NO, GAIN, GAIN_BLOCKS };
    static Ptr<ExposureCompensator> createDefault(int type);
    void feed(const std::vector<Point>& corners, const std::vector<UMat>& images, const std::vector<UMat>& masks);
    virtual void apply(int index, Point corner, InputOutputArray image, InputArray mask) = 0;
};
class BlocksGainCompensator : public ExposureCompensator {
  public:
    BlocksGainCompensator(int bl_width = 32, int bl_height = 32);
    void apply(int index, Point corner, InputOutputArray image, InputArray mask);
  private:
    int bl_width_, bl_height_;
    std::vector<UMat> gain_maps_;
};
}}
