// declaration-only stand-in (see tests/stubs/README.md)
#ifndef EBVO_STUB_CV_EIGEN_HPP
#define EBVO_STUB_CV_EIGEN_HPP
#include "../opencv.hpp"
#include <Eigen/Core>
namespace cv
{
template <class M> void eigen2cv(const M &, Mat &);
template <class M> void cv2eigen(const Mat &, M &);
} // namespace cv
#endif
