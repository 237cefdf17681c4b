// ndt_pose.h -- 4x4 pose arithmetic shared by the host and the device side of the fuser bank (csrc/ndt_fuser.hip,
// csrc/ndtgpu_api.hip): Eigen::Affine3d products, the rigid inverse and MatrixBase::eulerAngles(0, 1, 2) as the call sites of
// ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:124-126, 137-146, 415-466, 479 use them.  Matrices are 16
// doubles, COLUMN-major, exactly Eigen::Affine3d::data().  The same loops, in the same order and without fused
// multiply-adds, as the host mirror's stand-in types (host/ndt_gpu_types.h): a pose that went through the device is the pose
// the mirror computes.
#pragma once
#include "ndt_common.h"
#include <math.h>

// C = A * B
NDT_HD void ndt_pose_mul(const double *A, const double *B, double *C)
{
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += A[k * 4 + r] * B[c * 4 + k];
            C[c * 4 + r] = s;
        }
}

// the inverse of a rigid transform: R^T, -R^T t
NDT_HD void ndt_pose_inverse(const double *A, double *C)
{
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    for (int q = 0; q < 16; q++) C[q] = (q % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[j * 4 + i] = A[i * 4 + j];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int j = 0; j < 3; j++) s += C[j * 4 + i] * A[12 + j];
        C[12 + i] = -s;
    }
}

// rotation().eulerAngles(0, 1, 2) of a pose, Eigen 3.3's branches (the first angle lands in [0, pi]; host/ndt_gpu_types.h)
NDT_HD void ndt_euler012(const double *T, double *e)
{
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    auto R = [&](int r, int c) { return T[c * 4 + r]; };
    const double pi = 3.14159265358979323846;
    double res0 = atan2(R(1, 2), R(2, 2)), res1, res2;
    const double c2 = sqrt(R(0, 0) * R(0, 0) + R(0, 1) * R(0, 1));
    if (res0 > 0.0) {
        res0 -= pi;
        res1 = atan2(-R(0, 2), -c2);
    } else {
        res1 = atan2(-R(0, 2), c2);
    }
    const double s1 = sin(res0), c1 = cos(res0);
    res2 = atan2(s1 * R(2, 0) - c1 * R(1, 0), c1 * R(1, 1) - s1 * R(2, 1));
    e[0] = -res0; e[1] = -res1; e[2] = -res2;
}

// what the post-registration step of a fuser needs of NDTFeatureFuserHMT::Params (ndt_feature_fuser_hmt.h:58-207) and of the
// two fuse deltas the constructor sets (:222-223)
struct NdtFuserPolicy {
    double max_translation_norm, max_rotation_norm, translation_fuse_delta, rotation_fuse_delta;
    int check_consistency, fuse_incomplete, all_matches_valid, force_odom_as_est, compute_cov;
};

// a fuser's pose state, device resident (host copy: ndtgpu_fuser_bank)
struct NdtFuserState {
    double Tnow[16], Tlast_fuse[16];
    double cov_mean[3], cov[9];        // current_posecov: Pose2d mean, 3x3 covariance (column-major)
};

struct NdtFuserResultDev {             // mirrors ndtgpu_fuser_result
    double Tnow[16], Tmotion_est[16], spose[16];
    NdtMatchResultDev match;
    int32_t match_ok, registration_failure, cov_singular, pad_;
    double posecov_mean[3], posecov[9];
};

hipError_t ndt_launch_cloud_transform(const void *xyz_dev, size_t count, size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                      const double *T16a_dev, const double *T16b_dev, size_t T_stride_doubles, float *out_dev,
                                      hipStream_t stream);
hipError_t ndt_launch_fuser_post(const NdtFuserPolicy &pol, const double *sensor_pose16_dev, NdtFuserState *state_dev,
                                 const double *Tmotion16_dev, const double *Test16_dev, const NdtMatchResultDev *match_dev,
                                 const double *cov36_dev, const int *cov_singular_dev, size_t count, double *spose16_dev,
                                 double *fuse_origin3_dev, NdtFuserResultDev *out_dev, hipStream_t stream);
