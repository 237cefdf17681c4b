// ndt_gpu_types.h -- the two container seams the host mirror needs when it is built WITHOUT the
// reference's dependencies (no Eigen / PCL in this image).  In the real catkin build define
// NDTGPU_USE_EIGEN_PCL before including lslgeneric_gpu.h and these aliases become
// Eigen::Affine3d / pcl::PointXYZ / pcl::PointCloud -- the call sites do not change.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

#ifdef NDTGPU_USE_EIGEN_PCL
#include <Eigen/Geometry>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace ndtgpu_host {
using Affine3d = Eigen::Affine3d;
using PointXYZ = pcl::PointXYZ;
template <class P> using PointCloud = pcl::PointCloud<P>;
inline const double *affine_data(const Affine3d &T) { return T.data(); }
inline double *affine_data(Affine3d &T) { return T.data(); }
}  // namespace ndtgpu_host
#else
namespace ndtgpu_host {

// 4x4 homogeneous transform, COLUMN-major like Eigen::Affine3d::data()
struct Affine3d {
    double m[16];
    Affine3d() { setIdentity(); }
    void setIdentity()
    {
        for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
    static Affine3d Identity() { return Affine3d(); }
    double &operator()(int r, int c) { return m[c * 4 + r]; }
    double operator()(int r, int c) const { return m[c * 4 + r]; }
    double *data() { return m; }
    const double *data() const { return m; }
    Affine3d operator*(const Affine3d &o) const
    {
        Affine3d r;
        for (int c = 0; c < 4; c++)
            for (int rr = 0; rr < 4; rr++) {
                double s = 0;
                for (int k = 0; k < 4; k++) s += (*this)(rr, k) * o(k, c);
                r(rr, c) = s;
            }
        return r;
    }
    Affine3d inverse() const   // rigid inverse
    {
        Affine3d r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r(i, j) = (*this)(j, i);
        for (int i = 0; i < 3; i++) {
            double s = 0;
            for (int j = 0; j < 3; j++) s += r(i, j) * (*this)(j, 3);
            r(i, 3) = -s;
        }
        return r;
    }
    void translation(double t[3]) const { t[0] = m[12]; t[1] = m[13]; t[2] = m[14]; }
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
inline const double *affine_data(const Affine3d &T) { return T.data(); }
inline double *affine_data(Affine3d &T) { return T.data(); }

struct PointXYZ {   // pcl::PointXYZ layout: 16 bytes
    float x, y, z, pad;
    PointXYZ() : x(0), y(0), z(0), pad(1.f) {}
    PointXYZ(float X, float Y, float Z) : x(X), y(Y), z(Z), pad(1.f) {}
};

template <class P> struct PointCloud {
    std::vector<P> points;
    void push_back(const P &p) { points.push_back(p); }
    size_t size() const { return points.size(); }
    const P &front() const { return points.front(); }
    const P &back() const { return points.back(); }
};

}  // namespace ndtgpu_host
#endif
