// DECLARATIONS ONLY - see tests/mock_opencv/README.md.  The member layout follows OpenCV 3.4's stitching/detail/blenders.hpp
#pragma once
#include "../../core.hpp"
namespace cv { namespace detail {
class Blender {
  public:
    virtual ~Blender();
    enum { NO, FEATHER, MULTI_BAND };
    static Ptr<Blender> createDefault(int type, bool try_gpu = false);
    void prepare(const std::vector<Point>& corners, const std::vector<Size>& sizes);
    virtual void prepare(Rect dst_roi);
    virtual void feed(InputArray img, InputArray mask, Point tl);
    virtual void blend(InputOutputArray dst, InputOutputArray dst_mask);
  protected:
    UMat dst_, dst_mask_; Rect dst_roi_;
};
class MultiBandBlender : public Blender {
  public:
    MultiBandBlender(int try_gpu = false, int num_bands = 5, int weight_type = CV_32F);
    int numBands() const; void setNumBands(int);
  private:
    int actual_num_bands_, num_bands_;
    std::vector<UMat> dst_pyr_laplace_;
    std::vector<UMat> dst_band_weights_;
    Rect dst_roi_final_;
    bool can_use_gpu_; int weight_type_;
};
}}
