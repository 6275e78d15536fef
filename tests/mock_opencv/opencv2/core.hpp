// DECLARATIONS ONLY - see tests/mock_opencv/README.md.  For g++ -fsyntax-only of tools/opencv_pin/pin.cpp; pins nothing.
#pragma once
#include <cstddef>
#include <string>
#include <vector>
#define CV_VERSION "3.4.mock"
#define CV_8U 0
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
namespace cv {
typedef std::string String;
typedef unsigned char uchar;
struct Point { int x, y; Point(); Point(int, int); };
struct Size { int width, height; Size(); Size(int, int); int area() const; };
struct Rect { int x, y, width, height; Rect(); Rect(int, int, int, int); Point tl() const; Size size() const; };
struct Scalar { Scalar(); Scalar(double, double = 0, double = 0, double = 0); static Scalar all(double); };
enum { ACCESS_READ = 1 << 24 };
enum { BORDER_CONSTANT = 0, BORDER_REFLECT = 2 };
class Mat; class UMat; class MatExpr;
class _InputArray { public: _InputArray(); _InputArray(const Scalar&); _InputArray(const Mat&); _InputArray(const UMat&); _InputArray(const MatExpr&);
                    template <class T> _InputArray(const std::vector<T>&); };
class _OutputArray : public _InputArray { public: _OutputArray(); _OutputArray(Mat&); _OutputArray(UMat&); };
class _InputOutputArray : public _OutputArray { public: _InputOutputArray(); _InputOutputArray(Mat&); _InputOutputArray(UMat&); };
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _InputOutputArray& InputOutputArray;
class Mat {
  public:
    int rows, cols; uchar* data;
    Mat(); Mat(int rows, int cols, int type); Mat(Size, int type); Mat(Size, int type, const Scalar&); Mat(const MatExpr&);
    Mat& operator=(const MatExpr&);
    Mat clone() const; bool empty() const; bool isContinuous() const; Size size() const;
    int depth() const; int channels() const; size_t elemSize() const; size_t total() const;
    void convertTo(OutputArray, int rtype, double alpha = 1, double beta = 0) const;
    void copyTo(OutputArray) const;
    template <class T> T* ptr(int y = 0); template <class T> const T* ptr(int y = 0) const;
    Mat operator()(const Rect&) const;
};
class MatExpr { public: operator Mat() const; };
MatExpr operator&(const Mat&, const Mat&);
template <class T> class Mat_ : public Mat { public: Mat_(); Mat_(int rows, int cols); Mat_(const Mat&); T& operator()(int, int); const T& operator()(int, int) const; Mat_ clone() const; };
class UMat {
  public:
    UMat(); void create(Size, int type); UMat& setTo(InputArray value); Size size() const;
    void convertTo(OutputArray, int rtype, double alpha = 1, double beta = 0) const;
    void copyTo(OutputArray) const; Mat getMat(int flags) const;
};
template <class T> class Ptr { public: Ptr(); template <class U> Ptr(const Ptr<U>&); T* operator->() const; T* get() const; T& operator*() const; };
template <class T, class... A> Ptr<T> makePtr(const A&...);
const String& getBuildInformation();
void vconcat(InputArray, InputArray, OutputArray);
}  // namespace cv
