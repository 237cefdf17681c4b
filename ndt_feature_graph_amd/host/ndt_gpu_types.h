// ndt_gpu_types.h -- the Eigen / PCL names the reference's call sites use, for builds WITHOUT those libraries
// (this image has neither).  In the real catkin build define NDTGPU_USE_EIGEN_PCL before including the mirror
// headers and these names ARE Eigen::Affine3d / Vector3d / Matrix3d / MatrixXd and pcl::PointXYZ / PointCloud --
// the mirror's signatures are written with the Eigen:: / pcl:: spellings either way, so they are the reference's own.
// The stand-ins implement only what the call sites of the path use (ndt_feature_fuser_hmt.cpp:65-512,
// ndt_feature_graph.cpp:24-353, ndt_matcher_d2d_fusion.h:797-1155).
#pragma once
#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <vector>

#ifdef NDTGPU_USE_EIGEN_PCL
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#else
namespace Eigen {

struct Vector3d {
    double v[3];
    Vector3d() : v{0, 0, 0} {}
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    Vector3d operator+(const Vector3d &o) const { return Vector3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vector3d operator-(const Vector3d &o) const { return Vector3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vector3d operator*(double s) const { return Vector3d(v[0] * s, v[1] * s, v[2] * s); }
    double dot(const Vector3d &o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    double norm() const { return std::sqrt(dot(*this)); }
    void setZero() { v[0] = v[1] = v[2] = 0; }
    const double *data() const { return v; }
    double *data() { return v; }
    static Vector3d UnitX() { return Vector3d(1, 0, 0); }
};

// column-major like Eigen
struct Matrix3d {
    double m[9];
    Matrix3d() { setZero(); }
    double &operator()(int r, int c) { return m[c * 3 + r]; }
    double operator()(int r, int c) const { return m[c * 3 + r]; }
    void setZero() { for (double &x : m) x = 0; }
    void setIdentity() { setZero(); m[0] = m[4] = m[8] = 1; }
    Matrix3d transpose() const { Matrix3d t; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t(r, c) = (*this)(c, r); return t; }
    Matrix3d operator*(const Matrix3d &o) const
    {
        Matrix3d r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += (*this)(i, k) * o(k, j); r(i, j) = s; }
        return r;
    }
    Vector3d operator*(const Vector3d &x) const
    {
        return Vector3d((*this)(0, 0) * x[0] + (*this)(0, 1) * x[1] + (*this)(0, 2) * x[2],
                        (*this)(1, 0) * x[0] + (*this)(1, 1) * x[1] + (*this)(1, 2) * x[2],
                        (*this)(2, 0) * x[0] + (*this)(2, 1) * x[1] + (*this)(2, 2) * x[2]);
    }
    Matrix3d operator+(const Matrix3d &o) const { Matrix3d r; for (int k = 0; k < 9; k++) r.m[k] = m[k] + o.m[k]; return r; }
    // MatrixBase::eulerAngles(a0, a1, a2) as Eigen 3.3 computes it (Eigen/src/Geometry/EulerAngles.h), for three different
    // axes: the first angle lands in [0, pi] -- a rotation about z alone whose R(1,2) is +1e-17 instead of 0 comes back as
    // (pi, -pi, yaw - pi), which is what the fuser's relpose and rotation gates see upstream (fuser_hmt.cpp:124-126, 376,
    // 386; the case utils_affine_test.cpp:32-58 prints).  Reproduced branch for branch, not "fixed".
    Vector3d eulerAngles(int a0, int a1, int a2) const
    {
        const Matrix3d &R = *this;
        if (a0 == a2 || a0 == a1 || a1 == a2 || a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a2 < 0 || a2 > 2)
            throw std::invalid_argument("eulerAngles: three different axes");
        const int odd = ((a0 + 1) % 3 == a1) ? 0 : 1;
        const int i = a0, j = (a0 + 1 + odd) % 3, k = (a0 + 2 - odd) % 3;
        double res[3];
        res[0] = std::atan2(R(j, k), R(k, k));
        const double c2 = std::sqrt(R(i, i) * R(i, i) + R(i, j) * R(i, j));
        if ((odd && res[0] < 0.0) || (!odd && res[0] > 0.0)) {
            if (res[0] > 0.0) res[0] -= M_PI;
            else res[0] += M_PI;
            res[1] = std::atan2(-R(i, k), -c2);
        } else {
            res[1] = std::atan2(-R(i, k), c2);
        }
        const double s1 = std::sin(res[0]), c1 = std::cos(res[0]);
        res[2] = std::atan2(s1 * R(k, i) - c1 * R(j, i), c1 * R(j, j) - s1 * R(k, j));
        if (!odd) { res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2]; }
        return Vector3d(res[0], res[1], res[2]);
    }
    const double *data() const { return m; }
    double *data() { return m; }
};

// dynamic column-major matrix: what the call sites do with Eigen::MatrixXd (6x6 covariances, 6x1 gradients)
struct MatrixXd {
    std::vector<double> d;
    int r_ = 0, c_ = 0;
    MatrixXd() {}
    MatrixXd(int r, int c) : d((size_t)r * c, 0.0), r_(r), c_(c) {}
    int rows() const { return r_; }
    int cols() const { return c_; }
    void resize(int r, int c) { d.assign((size_t)r * c, 0.0); r_ = r; c_ = c; }
    double &operator()(int r, int c) { return d[(size_t)c * r_ + r]; }
    double operator()(int r, int c) const { return d[(size_t)c * r_ + r]; }
    double &operator()(int i) { return d[i]; }
    double operator()(int i) const { return d[i]; }
    void setZero() { for (double &x : d) x = 0; }
    void setIdentity() { setZero(); for (int i = 0; i < r_ && i < c_; i++) (*this)(i, i) = 1; }
    static MatrixXd Identity(int r, int c) { MatrixXd m(r, c); m.setIdentity(); return m; }
    double norm() const { double s = 0; for (double x : d) s += x * x; return std::sqrt(s); }
    MatrixXd operator+(const MatrixXd &o) const { MatrixXd m(r_, c_); for (size_t k = 0; k < d.size(); k++) m.d[k] = d[k] + o.d[k]; return m; }
    MatrixXd &operator+=(const MatrixXd &o) { for (size_t k = 0; k < d.size(); k++) d[k] += o.d[k]; return *this; }
    const double *data() const { return d.data(); }
    double *data() { return d.data(); }
};

// 4x4 homogeneous transform, COLUMN-major like Eigen::Affine3d::data()
struct Affine3d {
    double m[16];
    Affine3d() { setIdentity(); }
    void setIdentity() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0; }
    static Affine3d Identity() { return Affine3d(); }
    double &operator()(int r, int c) { return m[c * 4 + r]; }
    double operator()(int r, int c) const { return m[c * 4 + r]; }
    double *data() { return m; }
    const double *data() const { return m; }
    Affine3d operator*(const Affine3d &o) const
    {
        Affine3d r;
        for (int c = 0; c < 4; c++)
            for (int rr = 0; rr < 4; rr++) { double s = 0; for (int k = 0; k < 4; k++) s += (*this)(rr, k) * o(k, c); r(rr, c) = s; }
        return r;
    }
    Vector3d operator*(const Vector3d &p) const
    {
        return Vector3d(m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12], m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13],
                        m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14]);
    }
    Affine3d inverse() const   // rigid inverse
    {
        Affine3d r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r(i, j) = (*this)(j, i);
        for (int i = 0; i < 3; i++) { double s = 0; for (int j = 0; j < 3; j++) s += r(i, j) * (*this)(j, 3); r(i, 3) = -s; }
        return r;
    }
    Vector3d translation() const { return Vector3d(m[12], m[13], m[14]); }
    void setTranslation(const Vector3d &t) { m[12] = t[0]; m[13] = t[1]; m[14] = t[2]; }
    Matrix3d rotation() const { Matrix3d R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = (*this)(r, c); return R; }
    // Translation(x,y,z) * Rx * Ry * Rz  (ndt_matcher_d2d_fusion.h:1036-1039)
    static Affine3d fromPose(double x, double y, double z, double rx, double ry, double rz)
    {
        double cx = std::cos(rx), sx = std::sin(rx), cy = std::cos(ry), sy = std::sin(ry), cz = std::cos(rz), sz = std::sin(rz);
        Affine3d T;
        T(0, 0) = cy * cz;                T(0, 1) = -cy * sz;               T(0, 2) = sy;
        T(1, 0) = cx * sz + sx * sy * cz; T(1, 1) = cx * cz - sx * sy * sz; T(1, 2) = -sx * cy;
        T(2, 0) = sx * sz - cx * sy * cz; T(2, 1) = sx * cz + cx * sy * sz; T(2, 2) = cx * cy;
        T(0, 3) = x; T(1, 3) = y; T(2, 3) = z;
        return T;
    }
};

}  // namespace Eigen

namespace pcl {
struct PointXYZ {   // pcl::PointXYZ layout: 16 bytes
    float x, y, z, pad;
    PointXYZ() : x(0), y(0), z(0), pad(1.f) {}
    PointXYZ(float X, float Y, float Z) : x(X), y(Y), z(Z), pad(1.f) {}
};
template <class P> struct PointCloud {
    std::vector<P> points;
    unsigned width = 0, height = 1;
    void push_back(const P &p) { points.push_back(p); width = (unsigned)points.size(); }
    size_t size() const { return points.size(); }
    const P &front() const { return points.front(); }
    const P &back() const { return points.back(); }
};
}  // namespace pcl
#endif

namespace ndtgpu_host {
using Affine3d = Eigen::Affine3d;
using PointXYZ = pcl::PointXYZ;
template <class P> using PointCloud = pcl::PointCloud<P>;
inline const double *affine_data(const Affine3d &T) { return T.data(); }
inline double *affine_data(Affine3d &T) { return T.data(); }
// Translation(x,y,z) * Rx * Ry * Rz as a free function (works with real Eigen too)
inline Affine3d affine_from_pose(double x, double y, double z, double rx, double ry, double rz)
{
    double cx = std::cos(rx), sx = std::sin(rx), cy = std::cos(ry), sy = std::sin(ry), cz = std::cos(rz), sz = std::sin(rz);
    Affine3d T = Affine3d::Identity();
    double *m = T.data();
    m[0] = cy * cz;                m[4] = -cy * sz;               m[8] = sy;
    m[1] = cx * sz + sx * sy * cz; m[5] = cx * cz - sx * sy * sz; m[9] = -sx * cy;
    m[2] = sx * sz - cx * sy * cz; m[6] = sx * cz + cx * sy * sz; m[10] = cx * cy;
    m[12] = x; m[13] = y; m[14] = z;
    return T;
}
// lslgeneric::transformPointCloudInPlace(T, cloud) (fuser_hmt.cpp:74-75, 190)
inline void transformPointCloudInPlace(const Affine3d &T, PointCloud<PointXYZ> &pc)
{
    const double *m = T.data();
    for (auto &p : pc.points) {
        const double x = p.x, y = p.y, z = p.z;
        p.x = (float)(m[0] * x + m[4] * y + m[8] * z + m[12]);
        p.y = (float)(m[1] * x + m[5] * y + m[9] * z + m[13]);
        p.z = (float)(m[2] * x + m[6] * y + m[10] * z + m[14]);
    }
}
}  // namespace ndtgpu_host
