// host_demo.cpp -- the reference's call sites, written against the host mirror and run through the C-ABI:
//   A. the graph front door: NDTFeatureGraph::initialize / update (ndt_feature_graph.cpp:24-144) driving
//      NDTFeatureFuserHMT::initialize / update (ndt_feature_fuser_hmt.cpp:65-512) over a synthetic trajectory
//      (the corridor of ndt_odom_debug.cpp:94-119, closed by end walls), with the shipped parameter set;
//   B. the offline edge refinement (ndt_feature_graph_opt.cpp:131-160): all possible links -> NDT registration
//      (+ covariance + occupancy overlap score) -> gates;
//   C. the Newton loop of ndt_matcher_d2d_fusion.h:847-1121 RE-TYPED on the host against the mirror
//      (pseudoTransformNDT -> NDTCell* vector, derivativesNDT with MatrixXd, lineSearchMT, in-place cell transform,
//      delete) -- one ndtgpu_derivatives call per evaluation -- and compared with the device-resident matchFusion;
//   D. loadPointCloudCentroid, NDTMatcherD2D_2D, and the arguments the mirror must reject.
// Exit code 0 = every check passed.  Without a GPU: checks that the library fails loudly (NDTGPU_ERR_NO_DEVICE).
#include "ndt_feature_graph_gpu.h"
#include "ndt_map_msg_gpu.h"
extern "C" {
#include "../../oracle/ndt_oracle.h"      // test infrastructure only: the CPU restatement as checker
}

#include <cstdio>
#include <random>

using namespace ndt_feature;

static int g_fails = 0;
#define CHECK(cond, ...)                                                     \
    do {                                                                     \
        if (!(cond)) { std::printf("FAIL (%s:%d): ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); g_fails++; } \
    } while (0)

// a closed corridor (two wavy walls + two end walls) seen from `sensor_pose_world`; points in the sensor frame,
// ordered by bearing like a laser sweep
static pcl::PointCloud<pcl::PointXYZ> corridor_scan(const Eigen::Affine3d &sensor_pose_world, unsigned seed, int n_beams = 6000)
{
    std::mt19937 rng(seed);
    std::normal_distribution<double> nd(0.0, 0.03);
    std::uniform_real_distribution<double> uz(0.0, 0.02);
    pcl::PointCloud<pcl::PointXYZ> pc;
    const double ox = sensor_pose_world(0, 3), oy = sensor_pose_world(1, 3);
    const double yaw = std::atan2(sensor_pose_world(1, 0), sensor_pose_world(0, 0));
    for (int j = 0; j < n_beams; j++) {
        const double phi = -M_PI + 2.0 * M_PI * (j + 0.5) / n_beams, a = phi + yaw;
        const double dx = std::cos(a), dy = std::sin(a);
        // march along the beam until a wall is crossed (walls: y = +-(2 + 0.3 sin(0.9 x)), x = -9, x = 13)
        double r = 0.0, step = 0.02;
        for (; r < 40.0; r += step) {
            const double x = ox + r * dx, y = oy + r * dy;
            if (x < -9.0 || x > 13.0 || y > 2.0 + 0.3 * std::sin(0.9 * x) || y < -2.0 - 0.2 * std::cos(0.7 * x)) break;
        }
        r += nd(rng);
        if (r < 0.3 || r > 30.0) continue;
        pc.push_back(pcl::PointXYZ((float)(r * std::cos(phi)), (float)(r * std::sin(phi)), (float)uz(rng)));
    }
    return pc;
}

// (not distanceBetweenAffine3d: that one keeps upstream's unclamped acos, NaN for two poses that agree to the last bit)
static void pose_error(const Eigen::Affine3d &a, const Eigen::Affine3d &b, double &d, double &ang)
{
    const Eigen::Affine3d rel = a.inverse() * b;
    d = rel.translation().norm();
    ang = std::fabs(getRobustYawFromAffine3d(rel, true));
}

// ---- C: the reference's host loop, re-typed ------------------------------------------------------------------
namespace retyped {
// computeHessianMahalanobis / computeScoreMahalanobis / computeGradientMahalanobis (fusion.h:11-32)
static Eigen::MatrixXd computeHessianMahalanobis(const Eigen::MatrixXd &Q)
{
    Eigen::MatrixXd H(6, 6);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) H(i, j) = Q(j, i) + Q(i, j);
    return H;
}
static double computeScoreMahalanobis(const double X[6], const Eigen::MatrixXd &Q)
{
    double s = 0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) s += X[i] * Q(i, j) * X[j];
    return s;
}
static Eigen::MatrixXd computeGradientMahalanobis(const double X[6], const Eigen::MatrixXd &Q)
{
    Eigen::MatrixXd g(6, 1);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) g(i, 0) += (Q(i, j) + Q(j, i)) * X[j];
    return g;
}
// Eigen::SelfAdjointEigenSolver<6x6> eigenvalues + vectors (cyclic Jacobi) and Hessian.ldlt().solve: only here, for the
// re-typed loop (the product does this on the device)
static void eig6(const Eigen::MatrixXd &A, double ev[6], double V[6][6])
{
    double a[6][6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { a[i][j] = 0.5 * (A(i, j) + A(j, i)); V[i][j] = (i == j); }
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int i = 0; i < 6; i++)
            for (int j = i + 1; j < 6; j++) off += a[i][j] * a[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 6; p++)
            for (int q = p + 1; q < 6; q++) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; k++) { double x = a[k][p], y = a[k][q]; a[k][p] = c * x - s * y; a[k][q] = s * x + c * y; }
                for (int k = 0; k < 6; k++) { double x = a[p][k], y = a[q][k]; a[p][k] = c * x - s * y; a[q][k] = s * x + c * y; }
                for (int k = 0; k < 6; k++) { double x = V[k][p], y = V[k][q]; V[k][p] = c * x - s * y; V[k][q] = s * x + c * y; }
            }
    }
    for (int i = 0; i < 6; i++) ev[i] = a[i][i];
}
static void solve6(const Eigen::MatrixXd &A, const Eigen::MatrixXd &b, double x[6])   // Gaussian elimination, partial pivoting
{
    double a[6][7];
    for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) a[i][j] = A(i, j); a[i][6] = b(i, 0); }
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        for (int j = 0; j < 7; j++) std::swap(a[c][j], a[piv][j]);
        for (int r = c + 1; r < 6; r++) { double f = a[r][c] / a[c][c]; for (int j = c; j < 7; j++) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 5; i >= 0; i--) { double s = a[i][6]; for (int j = i + 1; j < 6; j++) s -= a[i][j] * x[j]; x[i] = s / a[i][i]; }
}

static int g_evals = 0;

// ndt_feature::matchFusion, fusion.h:797-1155, with useNDT = true, useFeat = false (the feature maps are empty),
// useTikhonovRegularization = false -- statement by statement against the mirror's lslgeneric:: API.
static bool matchFusion(lslgeneric::NDTMap &targetNDT, lslgeneric::NDTMap &sourceNDT, Eigen::Affine3d &T, const Eigen::MatrixXd &Tcov,
                        bool useInitialGuess, bool step_control, int ITR_MAX, int n_neighbours, double DELTA_SCORE, bool useSoftConstraints)
{
    lslgeneric::NDTMatcherD2D matcher_d2d;
    matcher_d2d.n_neighbours = n_neighbours;
    bool convergence = false;
    double score_best = 1.7976931348623157e308;
    int itr_ctr = 0;
    double step_size = 1;
    double pose_increment_v[6], pose_local_v[6] = {0, 0, 0, 0, 0, 0};
    Eigen::MatrixXd Hessian(6, 6), score_gradient(6, 1), Hessian_ndt(6, 6), score_gradient_ndt(6, 1);
    Eigen::Affine3d TR, Tbest;
    bool ret = true;
    if (!useInitialGuess) T.setIdentity();
    Tbest = T;
    std::vector<lslgeneric::NDTCell *> nextNDT = sourceNDT.pseudoTransformNDT(T);
    Eigen::MatrixXd Q(6, 6);
    {   // Q = Tcov.inverse()
        for (int c = 0; c < 6; c++) {
            Eigen::MatrixXd e(6, 1);
            e(c, 0) = 1.0;
            double col[6];
            solve6(Tcov, e, col);
            for (int r = 0; r < 6; r++) Q(r, c) = col[r];
        }
    }
    auto free_cells = [&]() { for (unsigned int i = 0; i < nextNDT.size(); i++) if (nextNDT[i] != NULL) delete nextNDT[i]; };
    while (!convergence) {
        TR.setIdentity();
        Hessian.setZero();
        score_gradient.setZero();
        double score_here = 0.;
        double score_here_ndt = matcher_d2d.derivativesNDT(nextNDT, targetNDT, score_gradient_ndt, Hessian_ndt, true);
        g_evals++;
        score_here += score_here_ndt;
        Hessian += Hessian_ndt;
        score_gradient += score_gradient_ndt;
        if (useSoftConstraints) {
            score_here += computeScoreMahalanobis(pose_local_v, Q);
            Hessian += computeHessianMahalanobis(Q);
            score_gradient += computeGradientMahalanobis(pose_local_v, Q);
        }
        Eigen::MatrixXd scg = score_gradient;
        if (score_here < score_best) { Tbest = T; score_best = score_here; }
        double evals[6], evecs[6][6];
        eig6(Hessian, evals, evecs);
        double minCoeff = evals[0], maxCoeff = evals[0];
        for (int i = 1; i < 6; i++) { minCoeff = std::fmin(minCoeff, evals[i]); maxCoeff = std::fmax(maxCoeff, evals[i]); }
        if (minCoeff < 0) {
            double regularizer = score_gradient.norm();
            regularizer = regularizer + minCoeff > 0 ? regularizer : 0.001 * maxCoeff - minCoeff;
            for (int i = 0; i < 6; i++) evals[i] += regularizer;
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) {
                    double s = 0;
                    for (int k = 0; k < 6; k++) s += evecs[i][k] * evals[k] * evecs[j][k];
                    Hessian(i, j) = s;
                }
        }
        if (score_gradient.norm() <= DELTA_SCORE) {
            if (score_here > score_best) T = Tbest;
            free_cells();
            return true;
        }
        solve6(Hessian, score_gradient, pose_increment_v);
        for (int i = 0; i < 6; i++) pose_increment_v[i] = -pose_increment_v[i];
        double dginit = 0;
        for (int i = 0; i < 6; i++) dginit += pose_increment_v[i] * scg(i, 0);
        if (dginit > 0) {
            if (score_here > score_best) T = Tbest;
            free_cells();
            return true;
        }
        if (step_control) {
            struct V6 { double *p; double &operator()(int i) { return p[i]; } } incr{pose_increment_v};
            double step_size_ndt = matcher_d2d.lineSearchMT(incr, nextNDT, targetNDT);
            step_size = std::fmax(step_size_ndt, 0.);   // step_size_feat == 0 (fusion.h:1018-1023)
        } else {
            step_size = 1;
        }
        double inorm = 0;
        for (int i = 0; i < 6; i++) { pose_increment_v[i] *= step_size; inorm += pose_increment_v[i] * pose_increment_v[i]; }
        inorm = std::sqrt(inorm);
        TR = ndtgpu_host::affine_from_pose(pose_increment_v[0], pose_increment_v[1], pose_increment_v[2], pose_increment_v[3],
                                          pose_increment_v[4], pose_increment_v[5]);
        T = TR * T;
        for (int i = 0; i < 6; i++) pose_local_v[i] += pose_increment_v[i];
        for (unsigned int i = 0; i < nextNDT.size(); i++) {
            Eigen::Vector3d meanC = nextNDT[i]->getMean();
            Eigen::Matrix3d covC = nextNDT[i]->getCov();
            meanC = TR * meanC;
            covC = TR.rotation() * covC * TR.rotation().transpose();
            nextNDT[i]->setMean(meanC);
            nextNDT[i]->setCov(covC);
        }
        if (itr_ctr > 0) convergence = (inorm < DELTA_SCORE);
        if (itr_ctr > ITR_MAX) { convergence = true; ret = false; }
        itr_ctr++;
    }
    double score_here = matcher_d2d.derivativesNDT(nextNDT, targetNDT, score_gradient_ndt, Hessian_ndt, false);
    g_evals++;
    if (useSoftConstraints) score_here += computeScoreMahalanobis(pose_local_v, Q);
    if (score_here > score_best) T = Tbest;
    free_cells();
    return ret;
}
}  // namespace retyped

// What the untouched iSAM layer does with the graph (optimizeGraphUsingISAM, ndt_offline_mapper.h:40-107), with the
// optimiser replaced by "chain the link transforms from node 0": it only ever sees the three interfaces.
static void rewrite_poses_through_the_interfaces(ndt_feature::NDTFeatureGraphInterface &graph)
{
    std::vector<Eigen::Affine3d> pose(graph.getNbNodes());
    pose[0] = graph.getNodeInterface(0).getPose();
    std::vector<bool> have(graph.getNbNodes(), false);
    have[0] = true;
    for (size_t pass = 0; pass < graph.getNbNodes(); pass++)
        for (size_t i = 0; i < graph.getNbLinks(); i++) {
            const ndt_feature::NDTFeatureLinkInterface &link = graph.getLinkInterface(i);
            if (have[link.getRefIdx()] && !have[link.getMovIdx()]) {
                pose[link.getMovIdx()] = pose[link.getRefIdx()] * link.getRelPose();
                have[link.getMovIdx()] = true;
            }
        }
    for (size_t i = 0; i < graph.getNbNodes(); i++)
        if (have[i]) graph.getNodeInterface(i).setPose(pose[i]);
}

template <class F> static bool throws_invalid(F f)
{
    try { f(); } catch (const ndtgpu_host::Error &e) { return e.status == NDTGPU_ERR_INVALID; }
    return false;
}

int main()
{
    NDTFeatureFuserHMT::Params fp;       // the shipped parameter set (gustav_laser_tf.launch:8-88; ndt_graph_offline.cpp:265-330)
    fp.resolution = 0.5; fp.map_size_x = 60; fp.map_size_y = 60; fp.map_size_z = 1.0; fp.sensor_range = 30;
    fp.useNDT = true; fp.useFeat = false; fp.useOdom = false;
    fp.neighbours = 2; fp.stepcontrol = true; fp.ITR_MAX = 30; fp.DELTA_SCORE = 1e-6;
    fp.globalTransf = true; fp.loadCentroid = true; fp.fusion2d = false;      // (the defaults: what the one-call device path covers)
    fp.useSoftConstraints = true; fp.useTikhonovRegularization = false; fp.computeCov = true;
    NDTFeatureGraph::Params gp;
    gp.newNodeTranslDist = 1.0;
    gp.maxNodes = 8;
    InterestPointVec no_pts;

    // ---- 0. (no device needed) Eigen's eulerAngles(0, 1, 2) as the fuser uses it: first angle in [0, pi], the angles
    //         rebuild the rotation; the cases of utils_affine_test.cpp:32-58 and the +-1e-17 quirk of a plane rotation
    {
        auto check_euler = [&](const Eigen::Affine3d &A, const char *what) {
            const Eigen::Vector3d e = A.rotation().eulerAngles(0, 1, 2);
            const Eigen::Affine3d B = ndtgpu_host::affine_from_pose(0, 0, 0, e[0], e[1], e[2]);
            double err = 0;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) err = std::max(err, std::fabs(A.rotation()(r, c) - B.rotation()(r, c)));
            CHECK(e[0] >= 0.0 && e[0] <= M_PI + 1e-15 && err < 1e-12, "eulerAngles %s: (%.6f %.6f %.6f) rebuilds R to %.2e", what, e[0], e[1], e[2], err);
            return e;
        };
        const double off = 0.001;
        check_euler(ndtgpu_host::affine_from_pose(2., 0.1, 0., -M_PI + off, M_PI + off, M_PI + 0.9), "K8 a");
        check_euler(ndtgpu_host::affine_from_pose(2., 0.1, 0., 0.00104309, -0.000766065, -3.11517), "K8 b");
        const Eigen::Vector3d flat = check_euler(ndtgpu_host::affine_from_pose(0, 0, 0, 0, 0, 0.3), "plane rotation");
        CHECK(std::fabs(flat[0]) < 1e-15 && std::fabs(flat[2] - 0.3) < 1e-15, "plane rotation: yaw %.17g", flat[2]);
        Eigen::Affine3d q = ndtgpu_host::affine_from_pose(0, 0, 0, 0, 0, 0.3);
        q.data()[9] = 1e-17;                 // R(1,2) = +1e-17: Eigen answers (pi, +-pi, yaw - pi)
        const Eigen::Vector3d eq = q.rotation().eulerAngles(0, 1, 2);
        CHECK(std::fabs(eq[0] - M_PI) < 1e-12 && std::fabs(std::fabs(eq[1]) - M_PI) < 1e-12 && std::fabs(eq[2] - (0.3 - M_PI)) < 1e-12,
              "eulerAngles branch of R(1,2) > 0: (%.6f %.6f %.6f)", eq[0], eq[1], eq[2]);
        MotionModel2d mm0;
        const Eigen::Matrix3d Rm = mm0.getPose2dCov(Pose2d(0.5, 0.0, 0.0)).cov;    // (heading 0: the rotation of the covariance is the identity)
        CHECK(Rm(0, 1) == 0.0 && Rm(1, 0) == 0.0 && Rm(0, 2) == 0.0 && Rm(2, 0) == 0.0 && Rm(1, 2) == 0.0 && Rm(2, 1) == 0.0 && Rm(0, 0) > 0,
              "MotionModel2d::getMeasurementCov: off-diagonals must be zero (motion_model.cpp:202 R.setZero())");
    }

    if (ndtgpu_device_count() < 1) {
        try {
            NDTFeatureGraph g(gp, fp);
            pcl::PointCloud<pcl::PointXYZ> pc = corridor_scan(Eigen::Affine3d::Identity(), 1, 200);
            g.initialize(Eigen::Affine3d::Identity(), pc, no_pts);
            std::printf("FAIL: graph initialisation succeeded without a device\n");
            return 1;
        } catch (const ndtgpu_host::Error &e) {
            if (e.status != NDTGPU_ERR_NO_DEVICE) { std::printf("FAIL: wrong status %d\n", e.status); return 1; }
            std::printf("no GPU: %s -- OK (no CPU fallback)\n", e.what());
            return 0;
        }
    }

    // ---- A. NDTFeatureGraph::initialize / update over a trajectory -----------------------------------------------
    NDTFeatureGraph graph(gp, fp);
    const int K = 14;
    std::vector<Eigen::Affine3d> gt;
    for (int k = 0; k < K; k++) gt.push_back(ndtgpu_host::affine_from_pose(-6.0 + 0.3 * k, 0.05 * std::sin(0.7 * k), 0, 0, 0, 0.015 * k));
    std::mt19937 rng(5);
    std::normal_distribution<double> odo_t(0.0, 0.01), odo_r(0.0, 0.002);
    {
        pcl::PointCloud<pcl::PointXYZ> pc = corridor_scan(gt[0], 100);
        graph.initialize(gt[0], pc, no_pts);
    }
    double worst_d = 0, worst_a = 0;
    for (int k = 1; k < K; k++) {
        Eigen::Affine3d inc = gt[k - 1].inverse() * gt[k];
        Eigen::Affine3d Tmotion = ndtgpu_host::affine_from_pose(inc(0, 3) + odo_t(rng), inc(1, 3) + odo_t(rng), 0, 0, 0,
                                                                std::atan2(inc(1, 0), inc(0, 0)) + odo_r(rng));
        pcl::PointCloud<pcl::PointXYZ> pc = corridor_scan(gt[k], 100 + k);
        Eigen::Affine3d Tnow = graph.update(Tmotion, pc, no_pts);
        double d, a;
        pose_error(gt[k], Tnow, d, a);
        worst_d = std::fmax(worst_d, d); worst_a = std::fmax(worst_a, a);
    }
    std::printf("A: %d scans -> %zu nodes; worst pose error along the trajectory %.4f m / %.5f rad\n", K, graph.getNbNodes(), worst_d, worst_a);
    CHECK(graph.getNbNodes() >= 3 && graph.getNbNodes() <= gp.maxNodes, "node count %zu", graph.getNbNodes());
    CHECK(worst_d < 0.05 && worst_a < 0.01, "trajectory drifted: %.4f m / %.5f rad", worst_d, worst_a);
    CHECK(graph.fullInit(), "a node is not initialised");
    for (size_t n = 0; n < graph.getNbNodes(); n++) {
        CHECK(graph.getMap((int)n)->numberOfActiveCells() > 30, "node %zu has %d cells", n, graph.getMap((int)n)->numberOfActiveCells());
        std::vector<float> occ = graph.getMap((int)n)->getOccupancy();
        size_t neg = 0, pos = 0;
        for (float o : occ) { neg += o < 0; pos += o > 0; }
        CHECK(neg > 200 && pos > 30, "node %zu: occupancy not carved (%zu free, %zu occupied cells)", n, neg, pos);
    }
    CHECK(graph.getNode(0).nbUpdates >= 2 && graph.getNode(0).getFuser().last_match.iterations > 0, "node 0 was not updated through matchFusion");

    // ---- B. all possible links -> NDT registration + covariance + overlap score -> gates -------------------------------------
    std::vector<NDTFeatureLink> links = graph.computeAllPossibleLinks();
    for (auto &l : links) CHECK(l.score >= 0.0 && l.score <= 1.0, "overlap score %g out of range", l.score);
    std::vector<NDTFeatureLink> serial = links;
    graph.updateLinksUsingNDTRegistration(links, 2, false);                   // three batched device calls
    for (auto &l : serial) graph.updateLinkUsingNDTRegistration(l, 2, false);    // the reference's loop shape
    for (size_t k = 0; k < links.size(); k++) {
        bool same = true;
        for (int q = 0; q < 16; q++) same = same && std::fabs(links[k].T.data()[q] - serial[k].T.data()[q]) < 1e-9;
        double asym = 0, tr = 0;
        for (int a = 0; a < 6; a++) { tr += links[k].cov_3d(a, a); for (int b = 0; b < 6; b++) asym = std::fmax(asym, std::fabs(links[k].cov_3d(a, b) - links[k].cov_3d(b, a))); }
        std::printf("B: link %zu-%zu iters %d converged %d score %.4f trace(cov) %.3e batch==single %d\n", links[k].ref_idx, links[k].mov_idx,
                    links[k].iterations, (int)links[k].converged, links[k].score, tr, (int)same);
        CHECK(same, "batched and single-link registration differ");
        CHECK(tr > 0 && asym < 1e-9 * tr, "covariance: trace %g asymmetry %g", tr, asym);
        CHECK(std::fabs(links[k].score - serial[k].score) < 1e-12, "scores differ");
    }
    std::vector<NDTFeatureLink> valid = graph.getValidLinks(links, 1.0, 3.0, 0.5, 1);
    CHECK(!valid.empty(), "no link survives the gates");
    // the graph through the abstract interfaces (interfaces.h:10-48), as ndt_offline_mapper.h:40 takes it: the refined
    // links rewrite the node poses; then back again
    {
        std::vector<Eigen::Affine3d> before;
        for (size_t i = 0; i < graph.getNbNodes(); i++) before.push_back(graph.getNode(i).T);
        graph.setLinks(links);
        rewrite_poses_through_the_interfaces(graph);
        double worst = 0;
        for (size_t i = 0; i < graph.getNbNodes(); i++) {
            double d, a;
            pose_error(graph.getNode(i).T, before[i], d, a);
            worst = std::fmax(worst, d);
        }
        std::ostringstream os;
        os << graph.getLinkInterface(0);
        std::printf("B: node poses rewritten through NDTFeatureGraphInterface from %zu registered links: they moved by at most %.3f m; link 0 prints as%s\n",
                    graph.getNbLinks(), worst, os.str().c_str());
        CHECK(worst < 0.1 && graph.getNbLinks() == links.size() && os.str().find("score") != std::string::npos, "interface consumer");
        for (size_t i = 0; i < graph.getNbNodes(); i++) graph.getNodeInterface(i).setPose(before[i]);
        graph.clearAllLinks();
    }

    // ---- C. the re-typed host loop vs the device-resident matchFusion -----------------------------------------------------
    {
        lslgeneric::NDTMap &target = *graph.getMap(0);
        lslgeneric::NDTMap src(new lslgeneric::LazyGrid(0.5), true);
        pcl::PointCloud<pcl::PointXYZ> pc = corridor_scan(gt[2], 777);
        src.guessSize(0, 0, 0, 30, 30, 1.0);
        src.loadPointCloud(pc, 30.0);
        src.computeNDTCells(lslgeneric::CELL_UPDATE_MODE_SAMPLE_VARIANCE);
        Eigen::Affine3d guess = graph.getNode(0).T.inverse() * gt[2];
        guess = ndtgpu_host::affine_from_pose(guess(0, 3) + 0.06, guess(1, 3) - 0.04, 0, 0, 0, std::atan2(guess(1, 0), guess(0, 0)) + 0.01);
        MotionModel2d mm;
        Eigen::MatrixXd Tcov = mm.getCovMatrix6(Pose2d(0.6, 0.05, 0.03));
        Tcov(2, 2) = 1; Tcov(3, 3) = 1; Tcov(4, 4) = 1;
        std::vector<std::pair<int, int> > corr;
        for (int soft = 0; soft < 2; soft++) {
            Eigen::Affine3d Th = guess, Td = guess;
            retyped::g_evals = 0;
            bool rh = retyped::matchFusion(target, src, Th, Tcov, true, true, 30, 2, 1e-6, soft != 0);
            ndtgpu_match_result res;
            lslgeneric::NDTMap feat_t(new lslgeneric::CellVector(), true), feat_s(new lslgeneric::CellVector(), true);   // empty feature maps
            bool rd = matchFusion(target, src, feat_t, feat_s, corr, Td, Tcov, true, true, false, true, 30, 2, 1e-6, soft != 0, true, false, &res);
            double d, a;
            pose_error(Th, Td, d, a);
            std::printf("C: soft=%d  host loop (%d Hessian evaluations) vs device loop (%d iterations): |dt| %.2e m |dyaw| %.2e rad  ret %d/%d\n", soft,
                        retyped::g_evals, res.iterations, d, a, (int)rh, (int)rd);
            CHECK(d < 1e-6 && a < 1e-6 && rh == rd, "re-typed host loop and device loop disagree");
            CHECK(retyped::g_evals == res.iterations + 1 || retyped::g_evals == res.iterations, "evaluation count %d vs iterations %d", retyped::g_evals, res.iterations);
            double dg, ag;
            pose_error(graph.getNode(0).T.inverse() * gt[2], Td, dg, ag);
            CHECK(dg < 0.05, "matchFusion is %.3f m from the true pose", dg);
        }
    }

    // ---- D. loadPointCloudCentroid, NDTMatcherD2D_2D, rejected arguments ---------------------------------------------------
    {
        lslgeneric::NDTMap &node0 = *graph.getMap(0);
        Eigen::Vector3d map_centroid;
        node0.getCentroid(map_centroid[0], map_centroid[1], map_centroid[2]);
        pcl::PointCloud<pcl::PointXYZ> pc = corridor_scan(gt[1], 901);
        ndtgpu_host::transformPointCloudInPlace(graph.getNode(0).T.inverse() * gt[1], pc);
        Eigen::Vector3d origin = (graph.getNode(0).T.inverse() * gt[1]).translation();
        lslgeneric::NDTMap local(new lslgeneric::LazyGrid(0.5), true);
        local.loadPointCloudCentroid(pc, origin, map_centroid, Eigen::Vector3d(31.5, 31.5, 1.0), 30.0);
        local.computeNDTCells(lslgeneric::CELL_UPDATE_MODE_SAMPLE_VARIANCE);
        double cx, cy, cz;
        local.getCentroid(cx, cy, cz);
        const double kx = (cx - map_centroid[0]) / 0.5, ky = (cy - map_centroid[1]) / 0.5;
        CHECK(std::fabs(kx - std::round(kx)) < 1e-12 && std::fabs(ky - std::round(ky)) < 1e-12, "centroid not on the node map's lattice");
        CHECK(local.numberOfActiveCells() > 30, "centroid-loaded map has %d cells", local.numberOfActiveCells());
        // cell faces coincide: every Gaussian cell of the local map has its centre on the node map's lattice too
        std::vector<lslgeneric::NDTCell *> cells = local.getAllCells();
        for (auto *c : cells) delete c;
        lslgeneric::NDTMatcherD2D_2D m2;
        m2.n_neighbours = 2; m2.ITR_MAX = 30; m2.DELTA_SCORE = 1e-6;
        Eigen::Affine3d T2 = ndtgpu_host::affine_from_pose(0.02, -0.01, 0, 0, 0, 0.0);
        m2.match(node0, local, T2, true);
        CHECK(T2(2, 3) == 0.0 && T2(2, 2) == 1.0 && std::sqrt(T2(0, 3) * T2(0, 3) + T2(1, 3) * T2(1, 3)) < 0.05, "2D matcher left the plane or diverged");
        CHECK(throws_invalid([&] { local.loadPointCloud(pc, 30.0); local.computeNDTCells(lslgeneric::CELL_UPDATE_MODE_COVARIANCE_INTERSECTION); }),
              "unsupported cell update mode accepted");
        CHECK(throws_invalid([&] { local.loadPointCloud(pc, 30.0); local.computeNDTCells(lslgeneric::CELL_UPDATE_MODE_SAMPLE_VARIANCE, 1e5, 100.f); }),
              "occupancy_limit the plain build cannot honour accepted");
        // a fuser with the reference's DEFAULT feature switches (useFeat = useOdom = true, ndt_feature_fuser_hmt.h:77-79):
        // no interest points -> no feature cells; the 40 odometry cell pairs join the registration
        // (fuser_hmt.cpp:322-334, fusion.h:858-871, 1013-1023)
        NDTFeatureFuserHMT::Params dflt = fp;
        dflt.useFeat = true; dflt.useOdom = true;
        NDTFeatureFuserHMT fuser(dflt);
        fuser.setSensorPose(Eigen::Affine3d::Identity());
        fuser.initialize(gt[0], corridor_scan(gt[0], 100), no_pts);
        Eigen::Affine3d Tm = gt[0].inverse() * gt[1];
        Eigen::Affine3d Tn = fuser.update(Tm, corridor_scan(gt[1], 101), no_pts);
        double dd, da;
        pose_error(Tn, gt[1], dd, da);
        std::printf("D: fuser with useFeat / useOdom defaults: %d iterations, |dt| %.3f m |dyaw| %.4f rad from the true pose\n", fuser.last_match.iterations, dd, da);
        CHECK(fuser.last_match.iterations > 0 && dd < 0.05 && da < 0.01, "fuser with odometry cells is off: %.3f m %.4f rad", dd, da);
    }
    // ---- E. the map wire format: toMessage / fromMessage round trip (ndtgraph_conversion.h:34-43, 129-158) -----------------
    {
        ndt_map::NDTMapMsg msg;
        CHECK(lslgeneric::toMessage(graph.getMap(1), msg, "/world"), "toMessage failed");
        CHECK(msg.header.frame_id == "/world" && msg.x_cell_size == 0.5 && msg.x_size == 60.0 && msg.z_size == 1.0, "message geometry");
        size_t n_gauss = 0;
        for (const auto &c : msg.cells) n_gauss += c.hasGaussian_;
        CHECK((int)n_gauss == graph.getMap(1)->numberOfActiveCells() && msg.cells.size() > n_gauss, "message cells: %zu Gaussians of %zu", n_gauss, msg.cells.size());
        lslgeneric::LazyGrid *lz = nullptr;
        lslgeneric::NDTMap *back = nullptr;
        std::string frame;
        CHECK(lslgeneric::fromMessage(lz, back, msg, frame, true) && back != nullptr, "fromMessage failed");
        CHECK(frame == "/world" && back->numberOfActiveCells() == (int)n_gauss, "round trip lost cells: %d vs %zu", back->numberOfActiveCells(), n_gauss);
        // the map that came back registers like the original
        lslgeneric::NDTMatcherD2D m;
        m.n_neighbours = 2; m.ITR_MAX = 30; m.DELTA_SCORE = 1e-6;
        Eigen::Affine3d Ta = ndtgpu_host::affine_from_pose(0.03, -0.02, 0, 0, 0, 0.004), Tb = Ta;
        m.match(*graph.getMap(0), *graph.getMap(1), Ta, true);
        m.match(*graph.getMap(0), *back, Tb, true);
        double d, a;
        pose_error(Ta, Tb, d, a);
        std::printf("E: message with %zu cells (%zu Gaussians); registration against the round-tripped map differs by %.2e m\n", msg.cells.size(), n_gauss, d);
        CHECK(d < 1e-9 && a < 1e-7, "round-tripped map registers differently: %g m %g rad", d, a);   // acos near 1: 1e-16 in R(0,0) is 1.5e-8 rad
        // GPU map -> message -> ORACLE map (test infrastructure, oracle/ndt_oracle.h): the Gaussians the message carries give
        // the CPU restatement the same derivatives as the GPU map gives the device path
        {
            std::vector<double> mean3, cov9;
            for (const auto &c : msg.cells)
                if (c.hasGaussian_) {
                    mean3.push_back(c.mean_x); mean3.push_back(c.mean_y); mean3.push_back(c.mean_z);
                    for (int k = 0; k < 9; k++) cov9.push_back(c.cov_matrix[k]);
                }
            const double cen[3] = {msg.x_cen, msg.y_cen, msg.z_cen}, ext[3] = {msg.x_size, msg.y_size, msg.z_size};
            oracle_map *om = oracle_map_create(msg.x_cell_size, cen, ext);
            CHECK(om && oracle_map_set_cells(om, mean3.data(), cov9.data(), mean3.size() / 3) == 0, "oracle map from the message");
            std::vector<lslgeneric::NDTCell *> src = graph.getMap(0)->pseudoTransformNDT(graph.getNode(1).T.inverse() * graph.getNode(0).T);
            std::vector<double> sm, sc;
            for (auto *c : src) {
                const Eigen::Vector3d mu = c->getMean();
                const Eigen::Matrix3d C = c->getCov();
                for (int a = 0; a < 3; a++) sm.push_back(mu(a));
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) sc.push_back(C(a, b));
            }
            double go[6], Ho[36];
            const double so = oracle_derivatives(om, sm.data(), sc.data(), src.size(), 2, 1, 1.0, 0.05, go, Ho);
            lslgeneric::NDTMatcherD2D md;
            md.n_neighbours = 2;
            Eigen::MatrixXd gg(6, 1), Hg(6, 6);
            const double sg = md.derivativesNDT(src, *graph.getMap(1), gg, Hg, true);
            double eg = 0, eh = 0, ng = 0, nh = 0;
            for (int a = 0; a < 6; a++) {
                eg = std::fmax(eg, std::fabs(gg(a, 0) - go[a])); ng = std::fmax(ng, std::fabs(go[a]));
                for (int b = 0; b < 6; b++) { eh = std::fmax(eh, std::fabs(Hg(a, b) - Ho[a * 6 + b])); nh = std::fmax(nh, std::fabs(Ho[a * 6 + b])); }
            }
            std::printf("E: message -> oracle map: score %.6f vs %.6f on the GPU map, gradient off by %.1e (of %.1e), Hessian by %.1e (of %.1e)\n", so, sg, eg, ng, eh, nh);
            CHECK(std::fabs(so - sg) < 1e-9 * std::fabs(so) && eg < 1e-9 * ng && eh < 1e-9 * nh && std::fabs(so) > 1.0, "oracle map from the message disagrees");
            for (auto *c : src) delete c;
            oracle_map_destroy(om);
        }
        std::vector<float> o1 = graph.getMap(1)->getOccupancy(), o2 = back->getOccupancy();
        double worst = 0;
        for (size_t k = 0; k < o1.size(); k++)
            if (std::fabs(o1[k]) < 10.f) worst = std::fmax(worst, std::fabs(o1[k] - o2[k]));
        CHECK(worst < 1e-3, "occupancy round trip off by %g", worst);
        delete back;
    }
    // ---- F. scans in, poses out for a batch of pairs in ONE call (ndtgpu_host::ScanRegistrar over ndtgpu_register_batch_host)
    //         against the reference's pair-at-a-time sequence loadPointCloud + computeNDTCells + NDTMatcherD2D::match -----------
    {
        const int P = 10;
        const double cen[3] = {0, 0, 0}, ext[3] = {60, 60, 1.0};
        std::vector<pcl::PointCloud<pcl::PointXYZ>> fixed, moving;
        std::vector<Eigen::Affine3d> T, T_one;
        std::mt19937 r2(77);
        std::normal_distribution<double> gt_t(0.0, 0.03), gt_r(0.0, 0.004);
        for (int k = 0; k < P; k++) {
            fixed.push_back(corridor_scan(gt[k], 300 + k, 5000 + 37 * k));           // (clouds of different lengths)
            moving.push_back(corridor_scan(gt[k + 1], 400 + k, 5200 - 41 * k));
            const Eigen::Affine3d inc = gt[k].inverse() * gt[k + 1];
            T.push_back(ndtgpu_host::affine_from_pose(inc(0, 3) + gt_t(r2), inc(1, 3) + gt_t(r2), 0, 0, 0, std::atan2(inc(1, 0), inc(0, 0)) + gt_r(r2)));
        }
        T_one = T;
        lslgeneric::NDTMatcherD2D md;
        md.n_neighbours = 2; md.ITR_MAX = 30; md.DELTA_SCORE = 1e-6;
        ndtgpu_host::ScanRegistrar reg(0.5, cen, ext, 4, 3);                           // sub-batches of 4 pairs: 4 + 4 + 2
        std::vector<ndtgpu_match_result> rr;
        std::vector<bool> ok = reg.match(md, fixed, moving, T, 30.0, true, &rr);
        double worst = 0, worst_gt = 0;
        int same_iters = 0;
        for (int k = 0; k < P; k++) {
            lslgeneric::NDTMap tg(new lslgeneric::LazyGrid(0.5)), sr(new lslgeneric::LazyGrid(0.5));
            tg.guessSize(0, 0, 0, ext[0], ext[1], ext[2]); sr.guessSize(0, 0, 0, ext[0], ext[1], ext[2]);
            tg.loadPointCloud(fixed[k], 30.0); tg.computeNDTCells();
            sr.loadPointCloud(moving[k], 30.0); sr.computeNDTCells();
            const bool c1 = md.match(tg, sr, T_one[k], true);
            double d, a, dg, ag;
            pose_error(T_one[k], T[k], d, a);
            pose_error(gt[k].inverse() * gt[k + 1], T[k], dg, ag);
            worst = std::fmax(worst, std::fmax(d, a)); worst_gt = std::fmax(worst_gt, dg);
            same_iters += md.last_result.iterations == rr[k].iterations;
            CHECK(c1 == ok[k] && rr[k].n_target == tg.numberOfActiveCells() && rr[k].n_source == sr.numberOfActiveCells(),
                  "pair %d: converged %d / %d, cells %d / %d vs %d / %d", k, (int)c1, (int)ok[k], rr[k].n_target, rr[k].n_source,
                  tg.numberOfActiveCells(), sr.numberOfActiveCells());
        }
        std::printf("F: %d scan pairs in one call: poses within %.2e of the pair-at-a-time sequence (%d of %d with its iteration count), "
                    "within %.3f m of the true motion\n", P, worst, same_iters, P, worst_gt);
        CHECK(worst < 1e-6 && same_iters >= P - 1 && worst_gt < 0.08, "the batch call disagrees with the pair-at-a-time sequence");
    }
    // ---- G. NDTFeatureFuserHMT::update as ONE device call == the sequence of lslgeneric calls it stands for ---------------
    // (the sequence a catkin build runs when the reference's own ndt_feature_fuser_hmt.cpp is compiled against lslgeneric_gpu.h:
    //  scan into the node frame, scan map on the node map's lattice, odometry cells, matchFusion, covariance, pose, fuse-in)
    {
        NDTFeatureFuserHMT::Params p = fp;
        p.useOdom = true; p.useSoftConstraints = true; p.useTikhonovRegularization = true; p.stepControlFusion = true;
        const Eigen::Affine3d sensor = ndtgpu_host::affine_from_pose(0.2, -0.03, 0, 0, 0, 0.01);
        NDTFeatureFuserHMT one(p);
        one.setSensorPose(sensor);
        const Eigen::Affine3d start = ndtgpu_host::affine_from_pose(0.05, -0.02, 0, 0, 0, 0.004);
        const pcl::PointCloud<pcl::PointXYZ> first = corridor_scan(gt[0] * sensor, 600);
        one.initialize(start, first, no_pts);
        // call by call: the node map
        lslgeneric::NDTMap node(new lslgeneric::LazyGrid(p.resolution));
        node.initialize(start.translation()(0), start.translation()(1), 0., p.map_size_x, p.map_size_y, p.map_size_z);
        {
            pcl::PointCloud<pcl::PointXYZ> c(first);
            ndtgpu_host::transformPointCloudInPlace(sensor, c);
            ndtgpu_host::transformPointCloudInPlace(start, c);
            const Eigen::Affine3d at = start * sensor;
            node.addPointCloud(at.translation(), c, 0.1, 100.0, 0.1);
            node.computeNDTCells(lslgeneric::CELL_UPDATE_MODE_SAMPLE_VARIANCE, 1e5, 255, at.translation(), 0.1);
        }
        CHECK(node.numberOfActiveCells() == one.map->numberOfActiveCells(), "G: node maps differ after initialize: %d / %d cells",
              node.numberOfActiveCells(), one.map->numberOfActiveCells());
        Eigen::Affine3d pose = start;
        const Eigen::Vector3d local_size(p.sensor_range + 3 * p.resolution, p.sensor_range + 3 * p.resolution, p.map_size_z);
        double worst = 0;
        int same_iters = 0;
        const int steps = 3;
        for (int k = 1; k <= steps; k++) {
            const Eigen::Affine3d inc = gt[k - 1].inverse() * gt[k];
            const Eigen::Affine3d Tmotion = ndtgpu_host::affine_from_pose(inc(0, 3) + 0.004 * k, inc(1, 3) - 0.003 * k, 0, 0, 0,
                                                                          std::atan2(inc(1, 0), inc(0, 0)) + 0.001 * k);
            const pcl::PointCloud<pcl::PointXYZ> scan = corridor_scan(gt[k] * sensor, 600 + k);
            const Eigen::Affine3d Ta = one.update(Tmotion, scan, no_pts);
            // -- the same scan, call by call
            const Eigen::Affine3d into_node = pose * sensor;
            pcl::PointCloud<pcl::PointXYZ> c(scan);
            ndtgpu_host::transformPointCloudInPlace(into_node, c);
            Eigen::Vector3d node_centre;
            node.getCentroid(node_centre[0], node_centre[1], node_centre[2]);
            lslgeneric::NDTMap scan_map(new lslgeneric::LazyGrid(p.resolution));
            scan_map.loadPointCloudCentroid(c, into_node.translation(), node_centre, local_size, p.sensor_range);
            scan_map.computeNDTCells(lslgeneric::CELL_UPDATE_MODE_SAMPLE_VARIANCE);
            MotionModel2d model;                                             // (default parameters, like `one`)
            const Pose2d rel(Tmotion.translation()[0], Tmotion.translation()[1], Tmotion.rotation().eulerAngles(0, 1, 2)[2]);
            Eigen::Matrix3d cell_cov = model.getPose2dCov(rel).cov;
            cell_cov(2, 2) = 0.01;
            Eigen::MatrixXd Tcov = model.getCovMatrix6(rel);
            Tcov(2, 2) = 1; Tcov(3, 3) = 1; Tcov(4, 4) = 1;
            lslgeneric::NDTCell was, is;                                     // the odometry cell pair: where the robot was + motion | where it is
            was.setMean(Tmotion.translation()); was.setCov(cell_cov);
            is.setMean(Eigen::Vector3d(0, 0, 0)); is.setCov(cell_cov);
            lslgeneric::NDTMap prev_local(new lslgeneric::CellVector(), true), curr_local(new lslgeneric::CellVector(), true);
            std::vector<std::pair<int, int> > corr;
            for (int i = 0; i < 40; i++) {
                addNDTCellToMap(&prev_local, &was);
                addNDTCellToMap(&curr_local, &is);
                corr.push_back(std::make_pair(i, i));
            }
            std::unique_ptr<lslgeneric::NDTMap> prev(prev_local.pseudoTransformNDTMap(pose)), curr(curr_local.pseudoTransformNDTMap(pose));
            curr->getMyIndex()->getCellIdx(39)->setCov(cell_cov);            // (the last one keeps the un-rotated covariance)
            Eigen::Affine3d est = Tmotion;
            ndtgpu_match_result mr;
            const bool ok = ndt_feature::matchFusion(node, scan_map, *prev, *curr, corr, est, Tcov, true, true, true, p.stepcontrol, p.ITR_MAX,
                                                     p.neighbours, p.DELTA_SCORE, true, true, true, &mr);
            pose = pose * (ok ? est : Tmotion);
            const Eigen::Affine3d at = pose * sensor;
            pcl::PointCloud<pcl::PointXYZ> raw(scan);
            ndtgpu_host::transformPointCloudInPlace(at, raw);
            node.addPointCloud(at.translation(), raw, 0.06, 25);
            node.computeNDTCells(lslgeneric::CELL_UPDATE_MODE_SAMPLE_VARIANCE, 1e5, 255, at.translation(), 0.1);
            double d = 0;
            for (int e = 0; e < 16; e++) d = std::fmax(d, std::fabs(Ta.data()[e] - pose.data()[e]));
            worst = std::fmax(worst, d);
            same_iters += mr.iterations == one.last_match.iterations && ok == (one.last_match.converged != 0);
            CHECK(node.numberOfActiveCells() == one.map->numberOfActiveCells(), "G: node maps differ after update %d: %d / %d cells", k,
                  node.numberOfActiveCells(), one.map->numberOfActiveCells());
        }
        std::printf("G: %d updates: the one-call fuser and the call-by-call sequence agree to %.2e on the pose (%d of %d with the same matcher report)\n",
                    steps, worst, same_iters, steps);
        CHECK(worst < 1e-9 && same_iters == steps, "the one-call fuser disagrees with the call-by-call sequence: %.3e", worst);
        // a configuration the device path does not cover is refused, not approximated
        NDTFeatureFuserHMT::Params q = fp;
        q.globalTransf = false;
        NDTFeatureFuserHMT other(q);
        CHECK(throws_invalid([&] { other.initialize(start, first, no_pts); }), "globalTransf = false accepted by the one-call fuser");
    }
    std::printf("%d failures in total\n", g_fails);
    return g_fails ? 1 : 0;
}
