// declaration-only stand-in (see tests/stubs/README.md)
#ifndef EBVO_STUB_OPENCV_HPP
#define EBVO_STUB_OPENCV_HPP
#include <cstddef>
#include <memory>
#include <string>
#include <vector>
typedef unsigned char uchar;
namespace cv
{
enum { CV_8U = 0, CV_8UC1 = 0, CV_8UC3 = 16, CV_32F = 5, CV_32FC1 = 5, CV_64F = 6, CV_64FC1 = 6, CV_16S = 3 };
enum { IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1, COLOR_HSV2BGR = 54, COLOR_GRAY2BGR = 8, COLOR_BGR2GRAY = 6, NORM_L2 = 4,
       BORDER_DEFAULT = 4, INTER_LINEAR = 1, LINE_AA = 16, FONT_HERSHEY_SIMPLEX = 0 };
template <class T> struct Point_
{
    T x, y;
    Point_();
    Point_(T, T);
    template <class U> Point_(const Point_<U> &);
    Point_ operator+(const Point_ &) const;
    Point_ operator-(const Point_ &) const;
    Point_ operator*(double) const;
    bool operator==(const Point_ &) const;
    bool operator!=(const Point_ &) const;
    double dot(const Point_ &) const;
};
typedef Point_<double> Point2d;
typedef Point_<float> Point2f;
typedef Point_<int> Point;
typedef Point_<int> Point2i;
template <class T> struct Point3_
{
    T x, y, z;
    Point3_();
    Point3_(T, T, T);
};
typedef Point3_<double> Point3d;
typedef Point3_<float> Point3f;
template <class T, int N> struct Vec
{
    T val[N];
    T &operator[](int);
    const T &operator[](int) const;
};
typedef Vec<uchar, 3> Vec3b;
typedef Vec<double, 3> Vec3d;
struct Scalar
{
    double val[4];
    Scalar();
    Scalar(double);
    Scalar(double, double, double, double = 0);
    double &operator[](int);
    const double &operator[](int) const;
};
struct Size
{
    int width, height;
    Size();
    Size(int, int);
};
struct Rect
{
    int x, y, width, height;
    Rect();
    Rect(int, int, int, int);
};
struct Range
{
    int start, end;
    Range(int, int);
    static Range all();
};
struct MatExpr;
struct Mat
{
    int rows, cols, flags, dims;
    uchar *data;
    struct Step
    {
        operator size_t() const;
        size_t operator[](int) const;
    } step;
    Mat();
    Mat(int, int, int);
    Mat(int, int, int, const Scalar &);
    Mat(int, int, int, void *, size_t = 0);
    Mat(Size, int);
    Mat(const Mat &);
    Mat(const MatExpr &);
    template <class T> Mat(const std::vector<T> &);
    Mat &operator=(const Mat &);
    Mat &operator=(const MatExpr &);
    Mat &operator=(const Scalar &);
    template <class T> T &at(int, int);
    template <class T> const T &at(int, int) const;
    template <class T> T &at(int);
    template <class T> const T &at(int) const;
    template <class T> T &at(Point_<int>);
    template <class T> T *ptr(int = 0);
    template <class T> const T *ptr(int = 0) const;
    uchar *ptr(int = 0);
    const uchar *ptr(int = 0) const;
    Mat clone() const;
    void copyTo(Mat &) const;
    void convertTo(Mat &, int, double = 1, double = 0) const;
    bool empty() const;
    int type() const;
    int channels() const;
    int depth() const;
    size_t total() const;
    size_t elemSize() const;
    bool isContinuous() const;
    Size size() const;
    Mat row(int) const;
    Mat col(int) const;
    Mat rowRange(int, int) const;
    Mat colRange(int, int) const;
    Mat reshape(int, int = 0) const;
    Mat t() const;
    Mat inv(int = 0) const;
    Mat mul(const Mat &, double = 1) const;
    double dot(const Mat &) const;
    Mat operator()(const Rect &) const;
    Mat operator()(Range, Range) const;
    void release();
    void push_back(const Mat &);
    static MatExpr zeros(int, int, int);
    static MatExpr zeros(Size, int);
    static MatExpr ones(int, int, int);
    static MatExpr eye(int, int, int);
};
struct MatExpr
{
    operator Mat() const;
    MatExpr mul(const Mat &, double = 1) const;
    MatExpr t() const;
    MatExpr inv(int = 0) const;
    double dot(const Mat &) const;
};
MatExpr operator+(const Mat &, const Mat &);
MatExpr operator-(const Mat &, const Mat &);
MatExpr operator*(const Mat &, const Mat &);
MatExpr operator/(const Mat &, const Mat &);
MatExpr operator+(const Mat &, const Scalar &);
MatExpr operator-(const Mat &, const Scalar &);
MatExpr operator-(const Mat &, double);
MatExpr operator+(const Mat &, double);
MatExpr operator*(const Mat &, double);
MatExpr operator*(double, const Mat &);
MatExpr operator/(const Mat &, double);
MatExpr operator-(const Mat &);
MatExpr operator+(const MatExpr &, const MatExpr &);
MatExpr operator-(const MatExpr &, const MatExpr &);
MatExpr operator*(const MatExpr &, const MatExpr &);
MatExpr operator*(const MatExpr &, double);
MatExpr operator/(const MatExpr &, double);
MatExpr operator*(const MatExpr &, const Mat &);
MatExpr operator*(const Mat &, const MatExpr &);
MatExpr operator-(const MatExpr &, const Scalar &);
template <class T> struct Mat_ : Mat
{
    Mat_();
    Mat_(int, int);
    Mat_(int, int, const T &);
    Mat_(const Mat &);
    Mat_(const MatExpr &);
    T &operator()(int, int);
    const T &operator()(int, int) const;
    T &operator()(int);
    const T &operator()(int) const;
    Mat_ &operator<<(const T &);
    Mat_ &operator,(const T &);
};
template <class T> struct Ptr : std::shared_ptr<T>
{
    Ptr();
    Ptr(T *);
    template <class U> Ptr(const std::shared_ptr<U> &);
};
struct KeyPoint
{
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint();
    KeyPoint(Point2f, float, float = -1, float = 0, int = 0, int = -1);
    KeyPoint(float, float, float, float = -1, float = 0, int = 0, int = -1);
};
struct DMatch
{
    int queryIdx, trainIdx, imgIdx;
    float distance;
};
struct Feature2D
{
    virtual ~Feature2D();
    virtual void compute(const Mat &, std::vector<KeyPoint> &, Mat &);
    virtual void detect(const Mat &, std::vector<KeyPoint> &, const Mat & = Mat());
    virtual void detectAndCompute(const Mat &, const Mat &, std::vector<KeyPoint> &, Mat &, bool = false);
};
struct SIFT : Feature2D
{
    static Ptr<SIFT> create(int = 0, int = 3, double = 0.04, double = 10, double = 1.6);
};
struct DescriptorMatcher
{
    virtual ~DescriptorMatcher();
    void knnMatch(const Mat &, const Mat &, std::vector<std::vector<DMatch>> &, int);
    void match(const Mat &, const Mat &, std::vector<DMatch> &);
};
struct BFMatcher : DescriptorMatcher
{
    BFMatcher(int = NORM_L2, bool = false);
    static Ptr<BFMatcher> create(int = NORM_L2, bool = false);
};
struct FlannBasedMatcher : DescriptorMatcher
{
    FlannBasedMatcher();
};
Scalar mean(const Mat &, const Mat & = Mat());
Scalar mean(const MatExpr &);
Scalar sum(const Mat &);
Scalar sum(const MatExpr &);
double norm(const Mat &, int = NORM_L2);
double norm(const Mat &, const Mat &, int = NORM_L2);
double norm(const MatExpr &, int = NORM_L2);
template <class T> double norm(const Point_<T> &);
template <class T> double norm(const Point3_<T> &);
void Sobel(const Mat &, Mat &, int, int, int, int = 3, double = 1, double = 0, int = BORDER_DEFAULT);
void GaussianBlur(const Mat &, Mat &, Size, double, double = 0, int = BORDER_DEFAULT);
void undistort(const Mat &, Mat &, const Mat &, const Mat &, const Mat & = Mat());
void cvtColor(const Mat &, Mat &, int, int = 0);
void hconcat(const Mat &, const Mat &, Mat &);
void hconcat(const std::vector<Mat> &, Mat &);
void vconcat(const Mat &, const Mat &, Mat &);
void buildPyramid(const Mat &, std::vector<Mat> &, int, int = BORDER_DEFAULT);
void triangulatePoints(const Mat &, const Mat &, const Mat &, const Mat &, Mat &);
void resize(const Mat &, Mat &, Size, double = 0, double = 0, int = INTER_LINEAR);
void sqrt(const Mat &, Mat &);
void pow(const Mat &, double, Mat &);
void multiply(const Mat &, const Mat &, Mat &, double = 1, int = -1);
void magnitude(const Mat &, const Mat &, Mat &);
void minMaxLoc(const Mat &, double *, double * = 0, Point * = 0, Point * = 0, const Mat & = Mat());
void normalize(const Mat &, Mat &, double = 1, double = 0, int = NORM_L2, int = -1, const Mat & = Mat());
Mat imread(const std::string &, int = IMREAD_COLOR);
bool imwrite(const std::string &, const Mat &, const std::vector<int> & = std::vector<int>());
void imshow(const std::string &, const Mat &);
int waitKey(int = 0);
void line(Mat &, Point, Point, const Scalar &, int = 1, int = 8, int = 0);
void circle(Mat &, Point, int, const Scalar &, int = 1, int = 8, int = 0);
void putText(Mat &, const std::string &, Point, int, double, Scalar, int = 1, int = 8, bool = false);
void rectangle(Mat &, Point, Point, const Scalar &, int = 1, int = 8, int = 0);
void drawMatches(const Mat &, const std::vector<KeyPoint> &, const Mat &, const std::vector<KeyPoint> &, const std::vector<DMatch> &,
                 Mat &);
template <class T> T saturate_cast(double);
namespace xfeatures2d
{
}
} // namespace cv
using cv::CV_16S;
using cv::CV_32F;
using cv::CV_32FC1;
using cv::CV_64F;
using cv::CV_64FC1;
using cv::CV_8U;
using cv::CV_8UC1;
using cv::CV_8UC3;
#endif
