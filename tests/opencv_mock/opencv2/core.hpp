// Test-only mock of the handful of OpenCV names afv_adapter.hpp touches under -DAFV_WITH_OPENCV (OpenCV is absent from
// the build image).  It exists so that the cv::Mat / cv::KeyPoint branches of the adapter are at least compiled; member
// names, types and the create(rows, cols, type) / ptr(row) / step / empty() signatures follow OpenCV 4.x core/mat.hpp and
// core/types.hpp.  Nothing links against it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#define CV_8U 0
typedef unsigned char uchar;
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
struct MatStep {
    size_t v = 0;
    operator size_t() const { return v; }
};
class Mat {
  public:
    int rows = 0, cols = 0;
    uchar *data = nullptr;
    MatStep step;
    void create(int r, int c, int /*type*/) { rows = r; cols = c; store.assign((size_t)r * c, 0); data = store.data(); step.v = (size_t)c; }
    uchar *ptr(int r = 0) { return data + (size_t)r * step.v; }
    const uchar *ptr(int r = 0) const { return data + (size_t)r * step.v; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  private:
    std::vector<uchar> store;
};
}  // namespace cv
