// ndt_fuser.hip -- the device side of ndtgpu_fuser_update_batch: what sits BETWEEN the grid build, the matcher and the
// ray-traced fuse-in when NDTFeatureFuserHMT::update runs for a batch of independent fusers without a host round trip.
//
// Replaces, per fuser slot (ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp):
//   :190          lslgeneric::transformPointCloudInPlace(Tinit_sensor_pose, cloud)       -> ndt_cloud_transform_kernel
//   :361-473      match_ok / allMatchesValid / computeCov / the consistency gate / the four ways Tnow is advanced /
//                 Tlast_fuse                                                             -> ndt_fuser_post_kernel
//   :479-480      spose = Tnow * sensor_pose, transformPointCloudInPlace(spose, cloud_orig) -> ndt_cloud_transform_kernel
// The kernels around them are the library's own: ndt_build*_kernel (loadPointCloudCentroid + computeNDTCells, :201-227),
// ndt_match_kernel (matchFusion / matchFusion2d, :353-357), ndt_covariance_kernel (:403-405), ndt_raytrace_kernel +
// ndt_fuse_finalize_kernel (addPointCloud + computeNDTCells, :485-486).
//
// HBM-bound (a cloud is read once and written once, 24 bytes per point); the pose arithmetic is 4x4 products on one
// thread per slot.  No fused multiply-add in either: the host mirror (host/ndt_gpu_types.h) rounds every product, and a
// transformed point must land in the same cell on both paths.
#include "ndt_pose.h"
#include "ndt_common.h"

namespace {

struct __attribute__((packed, aligned(4))) P3f { float x, y, z; };

}  // namespace

// out[k][i] = (float)(T[k] * in[k][i]) for the clouds of slots [0, count): T16 column-major, one workgroup column per cloud
// (blockIdx.y), records `stride` bytes apart, clouds `map_stride` bytes apart; the output is packed xyz (12 bytes).
// n_T: matrices applied in turn, EACH followed by its own rounding to float (fuser_hmt.cpp:74-75 transforms the first
// cloud twice: by the sensor pose, then by the initial pose).
__global__ __launch_bounds__(256) void ndt_cloud_transform_kernel(const char *__restrict__ in, size_t n_points, size_t stride,
                                                                   size_t map_stride, const double *__restrict__ T16a,
                                                                   const double *__restrict__ T16b, size_t T_stride_doubles,
                                                                   float *__restrict__ out)
{
#pragma clang fp contract(off)
    const size_t k = blockIdx.y;
    const double *A = T16a + k * T_stride_doubles, *Bm = T16b ? T16b + k * T_stride_doubles : nullptr;
    double a[12], b[12];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) { a[c * 3 + r] = A[c * 4 + r]; b[c * 3 + r] = Bm ? Bm[c * 4 + r] : 0.0; }
    const char *src = in + k * map_stride;
    P3f *dst = reinterpret_cast<P3f *>(out) + k * n_points;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_points; i += (size_t)gridDim.x * blockDim.x) {
        const P3f p = *reinterpret_cast<const P3f *>(src + i * stride);
        double x = (double)p.x, y = (double)p.y, z = (double)p.z;
        float fx = (float)(a[0] * x + a[3] * y + a[6] * z + a[9]);
        float fy = (float)(a[1] * x + a[4] * y + a[7] * z + a[10]);
        float fz = (float)(a[2] * x + a[5] * y + a[8] * z + a[11]);
        if (Bm) {
            x = (double)fx; y = (double)fy; z = (double)fz;
            fx = (float)(b[0] * x + b[3] * y + b[6] * z + b[9]);
            fy = (float)(b[1] * x + b[4] * y + b[7] * z + b[10]);
            fz = (float)(b[2] * x + b[5] * y + b[8] * z + b[11]);
        }
        dst[i] = P3f{fx, fy, fz};
    }
}

hipError_t ndt_launch_cloud_transform(const void *xyz_dev, size_t count, size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                      const double *T16a_dev, const double *T16b_dev, size_t T_stride_doubles, float *out_dev,
                                      hipStream_t stream)
{
    if (!count || !n_points) return hipSuccess;
    const unsigned bx = (unsigned)std::min<size_t>((n_points + 1023) / 1024, 64);     // 4 points per thread and pass at least
    hipLaunchKernelGGL(ndt_cloud_transform_kernel, dim3(bx, (unsigned)count), dim3(256), 0, stream, (const char *)xyz_dev, n_points,
                       stride_bytes, map_stride_bytes, T16a_dev, T16b_dev, T_stride_doubles, out_dev);
    return hipGetLastError();
}

// One thread per fuser slot: what NDTFeatureFuserHMT::update does between the registration and the fuse-in
// (fuser_hmt.cpp:361-480), on the device so that the fuse-in follows without the host.
__global__ __launch_bounds__(64) void ndt_fuser_post_kernel(NdtFuserPolicy pol, const double *__restrict__ sensor_pose16, NdtFuserState *__restrict__ state,
                                      const double *__restrict__ Tmotion16, const double *__restrict__ Test16,
                                      const NdtMatchResultDev *__restrict__ match, const double *__restrict__ cov36,
                                      const int *__restrict__ cov_singular, unsigned count, double *__restrict__ spose16,
                                      double *__restrict__ fuse_origin3, NdtFuserResultDev *__restrict__ out)
{
#pragma clang fp contract(off)
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    NdtFuserState &S = state[k];
    const double *Tm = Tmotion16 + 16 * (size_t)k, *Te = Test16 + 16 * (size_t)k;
    NdtFuserResultDev &R = out[k];
    R.match = match[k];
    // fuser_hmt.cpp:353-363
    bool match_ok = match[k].converged != 0 || pol.fuse_incomplete;
    if (pol.all_matches_valid) match_ok = true;
    int failure = 0, singular = 0;
    double Tnew[16];
    if (match_ok) {
        if (pol.compute_cov && cov36) {               // :399-413: the matcher's covariance joins the accumulated one
            const double *c6 = cov36 + 36 * (size_t)k;         // row-major 6x6
            singular = cov_singular ? cov_singular[k] : 0;
            double c3[9];                                      // cov6toCov3 (motion_model.cpp:141-152), column-major 3x3
            c3[0] = c6[0 * 6 + 0]; c3[4] = c6[1 * 6 + 1]; c3[8] = c6[5 * 6 + 5];
            // pose2dClearDependence (motion_model.cpp:154-165): the off-diagonal entries go
            c3[1] = c3[2] = c3[3] = c3[5] = c3[6] = c3[7] = 0.0;
            double e3[3];
            ndt_euler012(Te, e3);
            S.cov_mean[0] = Te[12]; S.cov_mean[1] = Te[13]; S.cov_mean[2] = e3[2];
            for (int q = 0; q < 9; q++) S.cov[q] = S.cov[q] + c3[q];
        }
        // :415-425: the registered increment against the odometry
        double Tinv[16], diff[16], e3[3];
        ndt_pose_inverse(Te, Tinv);
        ndt_pose_mul(Tinv, Tm, diff);
        ndt_euler012(diff, e3);
        const double dt = sqrt(diff[12] * diff[12] + diff[13] * diff[13] + diff[14] * diff[14]);
        const double dr = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
        if ((dt > pol.max_translation_norm || dr > pol.max_rotation_norm) && pol.check_consistency) {
            failure = 1;
            ndt_pose_mul(S.Tnow, Tm, Tnew);                    // "ALMOST DEFINATELY A REGISTRATION FAILURE": odometry
        } else {
            ndt_pose_mul(S.Tnow, pol.force_odom_as_est ? Tm : Te, Tnew);      // (globalTransf: Tnow * Tmotion_est)
            double Linv[16], df[16], ef[3];
            ndt_pose_inverse(S.Tlast_fuse, Linv);
            ndt_pose_mul(Linv, Tnew, df);
            ndt_euler012(df, ef);
            const double ft = sqrt(df[12] * df[12] + df[13] * df[13] + df[14] * df[14]);
            const double fr = sqrt(ef[0] * ef[0] + ef[1] * ef[1] + ef[2] * ef[2]);
            if (ft > pol.translation_fuse_delta || fr > pol.rotation_fuse_delta)
                for (int q = 0; q < 16; q++) S.Tlast_fuse[q] = Tnew[q];
        }
    } else {
        ndt_pose_mul(S.Tnow, Tm, Tnew);                        // :471-474
    }
    for (int q = 0; q < 16; q++) S.Tnow[q] = Tnew[q];
    // :479-480: where the sensor stood when the scan was taken, in the node map's frame
    double sp[16];
    ndt_pose_mul(Tnew, sensor_pose16, sp);
    for (int q = 0; q < 16; q++) { spose16[16 * (size_t)k + q] = sp[q]; R.spose[q] = sp[q]; R.Tnow[q] = Tnew[q]; R.Tmotion_est[q] = Te[q]; }
    fuse_origin3[3 * (size_t)k + 0] = sp[12]; fuse_origin3[3 * (size_t)k + 1] = sp[13]; fuse_origin3[3 * (size_t)k + 2] = sp[14];
    R.match_ok = match_ok ? 1 : 0;
    R.registration_failure = failure;
    R.cov_singular = singular;
    R.pad_ = 0;
    for (int q = 0; q < 3; q++) R.posecov_mean[q] = S.cov_mean[q];
    for (int q = 0; q < 9; q++) R.posecov[q] = S.cov[q];
}

hipError_t ndt_launch_fuser_post(const NdtFuserPolicy &pol, const double *sensor_pose16_dev, NdtFuserState *state_dev,
                                 const double *Tmotion16_dev, const double *Test16_dev, const NdtMatchResultDev *match_dev,
                                 const double *cov36_dev, const int *cov_singular_dev, size_t count, double *spose16_dev,
                                 double *fuse_origin3_dev, NdtFuserResultDev *out_dev, hipStream_t stream)
{
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(ndt_fuser_post_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, stream, pol, sensor_pose16_dev, state_dev,
                       Tmotion16_dev, Test16_dev, match_dev, cov36_dev, cov_singular_dev, (unsigned)count, spose16_dev, fuse_origin3_dev,
                       out_dev);
    return hipGetLastError();
}
