// ndt_feature_graph_gpu.h -- host front door of the ndt_feature classes that drive the hot path, over the C-ABI of libndtgpu.so:
//   ndt_feature::MotionModel2d        motion_model.hpp:123-163, motion_model.cpp:166-207
//   ndt_feature::matchFusion / matchFusion2d   ndt_matcher_d2d_fusion.h:797-1155, 1159-1176
//   ndt_feature::NDTFeatureFuserHMT   ndt_feature_fuser_hmt.h:36-334 (Params, initialize, update)
//   ndt_feature::NDTFeatureLink / NDTFeatureNode / overlapNDTOccupancyScore   ndt_feature_link.h:9-56, ndt_feature_node.h:38-252
//   ndt_feature::NDTFeatureGraph      ndt_feature_graph.h:20-280 (initialize, update, updateLink[s]UsingNDTRegistration, ...)
// Class, member and parameter names are the reference's, so that its callers compile against this header.  What is behind them
// is not the reference's control flow retyped but the batched device entries:
//   * NDTFeatureFuserHMT::initialize / update are ONE C-ABI call each (ndtgpu_fuser_initialize_batch / ndtgpu_fuser_update_batch
//     on the fuser's slot of a ndtgpu_host::FuserBank): scan -> scan map -> matchFusion -> covariance -> pose -> ray-traced
//     fuse-in run on the device without a host round trip.  A graph's node fusers are the slots of one bank over the pool its
//     node maps live in, so several graphs / robots / bags can also be stepped with one call (FuserBank::update).
//   * NDTFeatureGraph::updateLinksUsingNDTRegistration registers ALL links in one batched call each for the matcher, the
//     covariance and the overlap score (three calls per batch instead of three per link), never blocks on stdin
//     (graph.cpp:318-328) and reports non-convergence in NDTFeatureLink::converged.
// Configurations the device path does not cover are rejected loudly (ndtgpu_host::Error), never silently ignored: the FLIRT
// interest points (InterestPointVec is an empty placeholder: out of scope), globalTransf = false,
// loadCentroid = false.  In a catkin build that needs those the reference's own ndt_feature_fuser_hmt.cpp runs against
// lslgeneric_gpu.h instead.
#pragma once
#include "lslgeneric_gpu.h"

#include <cstdlib>
#include <iostream>
#include <sstream>

namespace ndt_feature {

struct InterestPointVec {};   // flirtlib's InterestPoint container: the feature front-end is out of scope (useFeat = false)

typedef Eigen::Vector3d Pose2d;
struct Pose2dCov {
    Pose2d mean;
    Eigen::Matrix3d cov;
};

// getRobustYawFromAffine3d (utils.h:30-41).  `clamped`: the cosine is brought into [-1, 1] first -- upstream does not, so a
// rotation whose (0, 0) entry is a rounding above 1 yields NaN there: kept as written where it only rejects a link
// (distanceBetweenAffine3d -> getValidLinks), not where it would turn a pose into NaN (force2D).
inline double getRobustYawFromAffine3d(const Eigen::Affine3d &a, bool clamped = true)
{
    const double c = clamped ? std::fmax(-1.0, std::fmin(1.0, a(0, 0))) : a(0, 0);   // x axis . rotated x axis, in the xy plane
    const double angle = std::acos(c);
    return (a(1, 0) > 0) ? angle : -angle;
}
// distanceBetweenAffine3d (utils.h:43-48)
inline void distanceBetweenAffine3d(const Eigen::Affine3d &p1, const Eigen::Affine3d &p2, double &dist, double &angularDist)
{
    Eigen::Affine3d tmp = p1.inverse() * p2;
    dist = tmp.translation().norm();
    angularDist = std::fabs(getRobustYawFromAffine3d(tmp, false));
}
// forceEigenAffine3dTo2dInPlace (utils.h:50-72)
inline void forceEigenAffine3dTo2dInPlace(Eigen::Affine3d &a3d)
{
    const double yaw = getRobustYawFromAffine3d(a3d);
    const Eigen::Vector3d t = a3d.translation();
    a3d = ndtgpu_host::affine_from_pose(t(0), t(1), 0., 0., 0., yaw);
}
// pose2dFromAffine3d, cov6toCov3, pose2dClearDependence (motion_model.cpp:128-165)
inline Pose2d pose2dFromAffine3d(const Eigen::Affine3d &T)
{
    return Pose2d(T.translation()(0), T.translation()(1), T.rotation().eulerAngles(0, 1, 2)(2));
}
inline Eigen::Matrix3d cov6toCov3(const Eigen::MatrixXd &cov6)
{
    Eigen::Matrix3d c;
    c(0, 0) = cov6(0, 0); c(1, 1) = cov6(1, 1); c(2, 2) = cov6(5, 5);
    c(0, 1) = cov6(0, 1); c(1, 0) = cov6(1, 0);
    c(0, 2) = cov6(0, 5); c(1, 2) = cov6(1, 5); c(2, 0) = cov6(5, 0); c(2, 1) = cov6(5, 1);
    return c;
}
inline void pose2dClearDependence(Pose2dCov &p)
{
    p.cov(0, 1) = p.cov(1, 0) = p.cov(0, 2) = p.cov(1, 2) = p.cov(2, 0) = p.cov(2, 1) = 0.;
}
// discardCell (utils.h:229-236)
inline bool discardCell(lslgeneric::NDTMap &map, const pcl::PointXYZ &pt) { return map.discardCellAtPoint(pt); }
// computeLocalCentroid (utils.h:213-227)
inline Eigen::Vector3d computeLocalCentroid(const Eigen::Vector3d &map_centroid, const Eigen::Vector3d &local_pos, double resolution)
{
    Eigen::Vector3d diff = map_centroid - local_pos, local_centroid;
    for (int i = 0; i < 3; i++) local_centroid(i) = diff(i) - std::floor(diff(i) / resolution) * resolution;
    return local_centroid;
}

// Eliazar-style odometry covariance (motion_model.hpp:123-163; motion_model.cpp:166-207)
class MotionModel2d {
public:
    class Params {
    public:
        Params() { Cd = 0.001; Ct = 0.001; Dd = 0.005; Dt = 0.005; Td = 0.001; Tt = 0.001; }
        double Cd, Ct, Dd, Dt, Td, Tt;
    };
    MotionModel2d() {}
    MotionModel2d(const MotionModel2d::Params &p) : params(p) {}
    void setParams(const MotionModel2d::Params &p) { params = p; }
    Pose2dCov getPose2dCov(const Pose2d &rel) const
    {
        Pose2dCov ret;
        ret.mean = rel;
        ret.cov = getMeasurementCov(rel);
        return ret;
    }
    Eigen::MatrixXd getCovMatrix6(const Pose2d &rel) const
    {
        Eigen::MatrixXd cov(6, 6);
        cov.setIdentity();
        Eigen::Matrix3d c2 = getMeasurementCov(rel);
        cov(0, 0) = c2(0, 0); cov(1, 0) = c2(1, 0); cov(0, 1) = c2(0, 1); cov(1, 1) = c2(1, 1);
        cov(0, 5) = c2(0, 2); cov(1, 5) = c2(1, 2);
        cov(5, 0) = c2(2, 0); cov(5, 1) = c2(2, 1); cov(5, 5) = c2(2, 2);
        return cov;
    }
    MotionModel2d::Params params;

private:
    Eigen::Matrix3d getMeasurementCov(const Eigen::Vector3d &rel) const
    {
        const double dist = std::sqrt(rel[0] * rel[0] + rel[1] * rel[1]), rot = rel[2];
        Eigen::Matrix3d R;
        R.setZero();                        // motion_model.cpp:202 (with the real Eigen an unset matrix is garbage)
        R(0, 0) = params.Dd * dist * dist + params.Dt * rot * rot;
        R(1, 1) = params.Cd * dist * dist + params.Ct * rot * rot;
        R(2, 2) = params.Td * dist * dist + params.Tt * rot * rot;
        return R;
    }
};

// ndt_feature::matchFusion (ndt_matcher_d2d_fusion.h:797-1155) -- the reference's full signature.  The whole Newton /
// More-Thuente loop runs on the device (ndtgpu_match_fusion_batch / ndtgpu_match_fusion_feat_batch); the feature maps
// are CellVector maps whose cells correspond through corr_feat (NDTMatcherFeatureD2D).
inline bool matchFusion(lslgeneric::NDTMap &targetNDT, lslgeneric::NDTMap &sourceNDT, lslgeneric::NDTMap &targetNDT_feat,
                        lslgeneric::NDTMap &sourceNDT_feat, const std::vector<std::pair<int, int> > &corr_feat,
                        Eigen::Affine3d &T, const Eigen::MatrixXd &Tcov, bool useInitialGuess, bool useNDT, bool useFeat,
                        bool step_control, int ITR_MAX = 30, int n_neighbours = 2, double DELTA_SCORE = 10e-4,
                        bool useSoftConstraints = true, bool step_control_fusion = true, bool useTikhonovRegularization = false,
                        ndtgpu_match_result *result = nullptr)
{
    if (!useNDT) throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "matchFusion: useNDT == false leaves nothing to match");
    lslgeneric::NDTMatcherD2D m;
    m.n_neighbours = n_neighbours;
    m.ITR_MAX = ITR_MAX;
    m.DELTA_SCORE = DELTA_SCORE;
    m.step_control = step_control;
    ndtgpu_match_params p = m.params(0x3f, useInitialGuess);
    ndtgpu_match_result r;
    uint32_t ti = (uint32_t)targetNDT.slot(), si = (uint32_t)sourceNDT.slot();
    double c36[36];
    const int flags = (useSoftConstraints ? 1 : 0) | (useTikhonovRegularization ? 2 : 0);
    if (flags) {
        if (Tcov.rows() != 6 || Tcov.cols() != 6) throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "matchFusion: Tcov must be 6x6");
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) c36[a * 6 + b] = Tcov(a, b);
    }
    // the feature / odometry-cell maps: NDTMatcherFeatureD2D(corr_feat) pairs source cell corr.second with target cell corr.first
    std::vector<double> sm, sc, tm, tc;
    uint32_t off[2] = {0, 0};
    if (useFeat) {
        lslgeneric::CellVector *cvt = targetNDT_feat.getMyIndex(), *cvs = sourceNDT_feat.getMyIndex();
        if (!cvt || !cvs) throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "matchFusion: the feature maps must be CellVector maps");
        for (const auto &c : corr_feat) {
            const lslgeneric::NDTCell *t = cvt->getCellIdx((unsigned)c.first), *s_ = cvs->getCellIdx((unsigned)c.second);
            if (!t || !s_) continue;                    // (targetNDT.getCellIdx fails: the pair is skipped)
            const Eigen::Vector3d mt = t->getMean(), ms = s_->getMean();
            const Eigen::Matrix3d Ct = t->getCov(), Cs = s_->getCov();
            for (int a = 0; a < 3; a++) { tm.push_back(mt(a)); sm.push_back(ms(a)); }
            const int ij[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
            for (int k = 0; k < 6; k++) { tc.push_back(Ct(ij[k][0], ij[k][1])); sc.push_back(Cs(ij[k][0], ij[k][1])); }
        }
        off[1] = (uint32_t)(tm.size() / 3);
    }
    if (off[1]) {
        ndtgpu_feat_pairs fp = {off, sm.data(), sc.data(), tm.data(), tc.data()};
        ndtgpu_host::check(ndtgpu_match_fusion_feat_batch(targetNDT.handle(), &ti, sourceNDT.handle(), &si, T.data(), flags ? c36 : nullptr, &fp,
                                                          1, &p, flags | (step_control_fusion ? 4 : 0), &r, nullptr),
                           "ndtgpu_match_fusion_feat_batch");
    } else {
        ndtgpu_host::check(ndtgpu_match_fusion_batch(targetNDT.handle(), &ti, sourceNDT.handle(), &si, T.data(), flags ? c36 : nullptr, 1,
                                                     &p, flags, &r, nullptr), "ndtgpu_match_fusion_batch");
    }
    if (result) *result = r;
    return r.converged != 0;
}
// ndt_feature::addNDTCellToMap (utils.h): a copy of the cell joins the map's cell list
inline void addNDTCellToMap(lslgeneric::NDTMap *map, lslgeneric::NDTCell *cell)
{
    lslgeneric::CellVector *cl = map->getMyIndex();
    if (!cl) return;
    lslgeneric::NDTCell *c = new lslgeneric::NDTCell();
    c->setMean(cell->getMean()); c->setCov(cell->getCov()); c->setN(cell->getN());
    cl->addCell(c);
}
// ndt_feature::matchFusion2d (ndt_matcher_d2d_fusion.h:1159-1176): NDTMatcherD2D_2D on the NDT maps
inline bool matchFusion2d(lslgeneric::NDTMap &targetNDT, lslgeneric::NDTMap &sourceNDT, lslgeneric::NDTMap & /*targetNDT_feat*/,
                          lslgeneric::NDTMap & /*sourceNDT_feat*/, const std::vector<std::pair<int, int> > & /*corr_feat*/,
                          Eigen::Affine3d &T, bool useInitialGuess, bool /*useNDT*/, bool /*useFeat*/, bool step_control, int ITR_MAX = 30,
                          int n_neighbours = 2, double DELTA_SCORE = 10e-4)
{
    lslgeneric::NDTMatcherD2D_2D matcher_d2d_2d;
    matcher_d2d_2d.n_neighbours = n_neighbours;
    matcher_d2d_2d.step_control = step_control;
    matcher_d2d_2d.ITR_MAX = ITR_MAX;
    matcher_d2d_2d.DELTA_SCORE = DELTA_SCORE;
    return matcher_d2d_2d.match(targetNDT, sourceNDT, T, useInitialGuess);
}

}  // namespace ndt_feature

namespace ndtgpu_host {

// B independent fusers stepped by ONE call (ndtgpu_fuser_bank): the slots' node maps are the maps of a MapPool.
class FuserBank {
public:
    FuserBank(const ndtgpu_fuser_params &p, std::shared_ptr<MapPool> pool) : pool_(std::move(pool)), prm_(p)
    {
        check(ndtgpu_fuser_bank_create(&p, pool_->size(), pool_->handle(), &bank_), "ndtgpu_fuser_bank_create");
    }
    ~FuserBank() { ndtgpu_fuser_bank_destroy(bank_); }
    FuserBank(const FuserBank &) = delete;
    FuserBank &operator=(const FuserBank &) = delete;
    const ndtgpu_fuser_params &params() const { return prm_; }
    const std::shared_ptr<MapPool> &pool() const { return pool_; }

    // slots [first, first + clouds.size()): poses and first clouds (sensor frame)
    void initialize(size_t first, const std::vector<Eigen::Affine3d> &initPos, const std::vector<const pcl::PointCloud<pcl::PointXYZ> *> &clouds)
    {
        pack(clouds);
        std::vector<double> T = flat(initPos);
        check(ndtgpu_fuser_initialize_batch_host(bank_, first, clouds.size(), T.data(), pts_.data(), np_, 16, np_ * 16), "ndtgpu_fuser_initialize_batch_host");
    }
    // one update of every slot of the range; returns the records of the call (pose, registered increment, matcher report)
    std::vector<ndtgpu_fuser_result> update(size_t first, const std::vector<Eigen::Affine3d> &Tmotion,
                                            const std::vector<const pcl::PointCloud<pcl::PointXYZ> *> &clouds, bool updateNDTMap = true)
    {
        if (Tmotion.size() != clouds.size()) throw Error(NDTGPU_ERR_INVALID, "FuserBank::update: one odometry increment per cloud");
        pack(clouds);
        std::vector<double> T = flat(Tmotion);
        check(ndtgpu_fuser_update_batch_host(bank_, first, clouds.size(), T.data(), pts_.data(), np_, 16, np_ * 16, updateNDTMap ? 1 : 0),
              "ndtgpu_fuser_update_batch_host");
        std::vector<ndtgpu_fuser_result> res(clouds.size());
        std::vector<double> now(16 * clouds.size());
        check(ndtgpu_fuser_poses(bank_, first, clouds.size(), now.data(), res.data()), "ndtgpu_fuser_poses");
        return res;
    }
    Eigen::Affine3d pose(size_t slot)
    {
        Eigen::Affine3d T;
        check(ndtgpu_fuser_poses(bank_, slot, 1, T.data(), nullptr), "ndtgpu_fuser_poses");
        return T;
    }

private:
    static std::vector<double> flat(const std::vector<Eigen::Affine3d> &T)
    {
        std::vector<double> o(16 * T.size());
        for (size_t k = 0; k < T.size(); k++)
            for (int e = 0; e < 16; e++) o[16 * k + e] = T[k].data()[e];
        return o;
    }
    // clouds of unequal length side by side, padded with NaN points (which the binning drops)
    void pack(const std::vector<const pcl::PointCloud<pcl::PointXYZ> *> &clouds)
    {
        static_assert(sizeof(pcl::PointXYZ) == 16, "pcl::PointXYZ is four floats");
        np_ = 0;
        for (const auto *c : clouds) np_ = std::max(np_, c->size());
        pts_.assign(clouds.size() * np_ * 4, std::nanf(""));
        for (size_t k = 0; k < clouds.size(); k++)
            for (size_t i = 0; i < clouds[k]->size(); i++) {
                const pcl::PointXYZ &q = clouds[k]->points[i];
                float *o = &pts_[(k * np_ + i) * 4];
                o[0] = q.x; o[1] = q.y; o[2] = q.z;
            }
    }
    std::shared_ptr<MapPool> pool_;
    ndtgpu_fuser_params prm_;
    ndtgpu_fuser_bank *bank_ = nullptr;
    std::vector<float> pts_;
    size_t np_ = 0;
};

}  // namespace ndtgpu_host

namespace ndt_feature {

class NDTFeatureFuserHMT {
public:
    Eigen::Affine3d Tnow, Tlast_fuse, Todom;   ///< current pose
    lslgeneric::NDTMap *map;                   ///< da map

    //! ndt_feature_fuser_hmt.h:58-207: all 30 fields, the reference's defaults
    class Params {
    public:
        Params()
        {
            checkConsistency = false;
            resolution = 1.;
            map_size_x = 40.; map_size_y = 40.; map_size_z = 10.;
            sensor_range = 3.;
            max_translation_norm = 1;
            max_rotation_norm = M_PI / 4.;
            fuseIncomplete = false;
            beHMT = true;
            prefix = "";
            hmt_map_dir = "map";
            useNDT = true; useFeat = true; useOdom = true;
            neighbours = 0;
            stepcontrol = true;
            ITR_MAX = 30;
            DELTA_SCORE = 10e-4;
            globalTransf = true;
            loadCentroid = true;
            forceOdomAsEst = false;
            visualizeLocalCloud = false;
            fusion2d = false;
            allMatchesValid = false;
            discardCells = false;
            useSoftConstraints = true;
            computeCov = true;
            stepControlFusion = true;
            useTikhonovRegularization = true;
        }
        bool checkConsistency;
        double resolution, map_size_x, map_size_y, map_size_z, sensor_range, max_translation_norm, max_rotation_norm;
        bool fuseIncomplete, beHMT;
        std::string prefix, hmt_map_dir;
        bool useNDT, useFeat, useOdom;
        int neighbours;
        bool stepcontrol;
        int ITR_MAX;
        double DELTA_SCORE;
        bool globalTransf, loadCentroid, forceOdomAsEst, visualizeLocalCloud, fusion2d, allMatchesValid, discardCells,
            useSoftConstraints, computeCov, stepControlFusion, useTikhonovRegularization;
        std::string getDescString() const
        {
            std::ostringstream os;
            os << "resolution" << resolution << "loadCentroid" << loadCentroid << "discardCells" << discardCells << "neighbours" << neighbours
               << "forceOdomAsEst" << forceOdomAsEst << "useSoftConstraints" << useSoftConstraints;
            return os.str();
        }
    };

    Params params_;
    MotionModel2d::Params motion_params_;
    ndtgpu_match_result last_match{};        // (added) what the device matcher reported for the last update
    ndtgpu_fuser_result last_update{};       // (added) the whole record of the last update

    // the fields of Params / MotionModel2d::Params / the sensor pose that the device path reads, as the C-ABI wants them.
    // Throws for the configurations it does not cover.
    static ndtgpu_fuser_params bankParams(const Params &p, const MotionModel2d::Params &m, const Eigen::Affine3d &sensor_pose)
    {
        if (!p.globalTransf || !p.loadCentroid)
            throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTFeatureFuserHMT: the device path covers globalTransf = loadCentroid = true (the "
                                                         "defaults); run the reference's own update against lslgeneric_gpu.h for the others");
        if (!p.useNDT) throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTFeatureFuserHMT: useNDT == false leaves nothing to match");
        ndtgpu_fuser_params q;
        ndtgpu_default_fuser_params(&q);
        q.resolution = p.resolution;
        q.map_size_x = p.map_size_x; q.map_size_y = p.map_size_y; q.map_size_z = p.map_size_z;
        q.sensor_range = p.sensor_range;
        q.max_translation_norm = p.max_translation_norm; q.max_rotation_norm = p.max_rotation_norm;
        q.check_consistency = p.checkConsistency; q.fuse_incomplete = p.fuseIncomplete;
        q.use_odom = p.useOdom;               // (useFeat: without FLIRT interest points there are no feature matches to add)
        q.neighbours = p.neighbours; q.stepcontrol = p.stepcontrol; q.itr_max = p.ITR_MAX; q.delta_score = p.DELTA_SCORE;
        q.force_odom_as_est = p.forceOdomAsEst; q.fusion2d = p.fusion2d; q.all_matches_valid = p.allMatchesValid;
        q.use_soft_constraints = p.useSoftConstraints; q.compute_cov = p.computeCov; q.step_control_fusion = p.stepControlFusion;
        q.use_tikhonov = p.useTikhonovRegularization; q.discard_cells = p.discardCells;
        q.motion_Cd = m.Cd; q.motion_Ct = m.Ct; q.motion_Dd = m.Dd; q.motion_Dt = m.Dt; q.motion_Td = m.Td; q.motion_Tt = m.Tt;
        for (int e = 0; e < 16; e++) q.sensor_pose[e] = sensor_pose.data()[e];
        return q;
    }

    NDTFeatureFuserHMT(const NDTFeatureFuserHMT::Params &params) : map(NULL), params_(params), isInit(false) {}
    // a fuser that is slot `slot` of a bank shared with other fusers (the node fusers of a graph)
    NDTFeatureFuserHMT(const NDTFeatureFuserHMT::Params &params, std::shared_ptr<ndtgpu_host::FuserBank> bank, size_t slot)
        : map(NULL), params_(params), isInit(false), bank_(std::move(bank)), slot_(slot) {}
    ~NDTFeatureFuserHMT()
    {
        if (map != NULL) delete map;
    }
    NDTFeatureFuserHMT(const NDTFeatureFuserHMT &) = delete;
    NDTFeatureFuserHMT &operator=(const NDTFeatureFuserHMT &) = delete;

    void setMotionParams(const MotionModel2d::Params &params) { motion_params_ = params; }
    void setSensorPose(Eigen::Affine3d spose) { sensor_pose = spose; }
    bool wasInit() { return isInit; }
    const NDTFeatureFuserHMT::Params &getParam() const { return params_; }
    Eigen::Matrix3d &getCov() { return current_posecov.cov; }

    /** initialize(initPos, cloud, pts, preLoad) -- ndt_feature_fuser_hmt.cpp:65-102 -- as ONE device call */
    void initialize(Eigen::Affine3d initPos, const pcl::PointCloud<pcl::PointXYZ> &cloudOrig, const InterestPointVec & /*pts*/, bool /*preLoad*/ = false)
    {
        if (!bank_) {     // a fuser on its own: a bank of one over a pool of one
            const double c[3] = {0, 0, 0}, sz[3] = {params_.map_size_x, params_.map_size_y, params_.map_size_z};
            auto pool = std::make_shared<ndtgpu_host::MapPool>(params_.resolution, c, sz, 1);
            bank_ = std::make_shared<ndtgpu_host::FuserBank>(bankParams(params_, motion_params_, sensor_pose), pool);
            slot_ = 0;
        }
        if (map != NULL) delete map;
        map = new lslgeneric::NDTMap(bank_->pool(), slot_);
        // (the host object learns where the map sits; the device call below fills it)
        map->initialize(initPos.translation()(0), initPos.translation()(1), 0., params_.map_size_x, params_.map_size_y, params_.map_size_z);
        bank_->initialize(slot_, {initPos}, {&cloudOrig});
        Tnow = Tlast_fuse = Todom = initPos;
        isInit = true;
    }

    /** update(Tmotion, cloud, pts, updateFeatureMap, updateNDTMap) -- ndt_feature_fuser_hmt.cpp:108-512 -- as ONE device call */
    Eigen::Affine3d update(Eigen::Affine3d Tmotion, const pcl::PointCloud<pcl::PointXYZ> &cloudOrig, const InterestPointVec & /*pts*/,
                           bool /*updateFeatureMap*/ = true, bool updateNDTMap = true)
    {
        if (!isInit) {
            fprintf(stderr, "NDT-FuserHMT: Call Initialize first!!\n");
            return Tnow;
        }
        absorb(bank_->update(slot_, {Tmotion}, {&cloudOrig}, updateNDTMap)[0], Tmotion);
        return Tnow;
    }
    // (added) what a batched caller does after FuserBank::update on this fuser's slot
    void absorb(const ndtgpu_fuser_result &r, const Eigen::Affine3d &Tmotion)
    {
        last_update = r;
        last_match = r.match;
        for (int e = 0; e < 16; e++) Tnow.data()[e] = r.Tnow[e];
        Todom = Todom * Tmotion;
        for (int a = 0; a < 3; a++) current_posecov.mean(a) = r.posecov_mean[a];
        for (int e = 0; e < 9; e++) current_posecov.cov.data()[e] = r.posecov[e];
    }
    const std::shared_ptr<ndtgpu_host::FuserBank> &bank() const { return bank_; }
    size_t slot() const { return slot_; }

private:
    bool isInit;
    Eigen::Affine3d sensor_pose;
    Pose2dCov current_posecov;
    std::shared_ptr<ndtgpu_host::FuserBank> bank_;
    size_t slot_ = 0;
};

// interfaces.h:10-48 -- what the iSAM layer consumes (ndt_offline_mapper.h:40 takes NDTFeatureGraphInterface&)
class NDTFeatureNodeInterface {
public:
    virtual ~NDTFeatureNodeInterface() {}
    virtual const Eigen::Affine3d &getPose() const = 0;
    virtual const Eigen::Matrix3d &getCov() const = 0;
    virtual void setPose(const Eigen::Affine3d &pose) = 0;
    virtual void setCov(const Eigen::Matrix3d &cov) = 0;
};
class NDTFeatureLinkInterface {
public:
    virtual ~NDTFeatureLinkInterface() {}
    virtual const Eigen::Affine3d &getRelPose() const = 0;
    virtual const Eigen::Matrix3d &getRelCov() const = 0;
    virtual double getScore() const = 0;
    virtual size_t getRefIdx() const = 0;
    virtual size_t getMovIdx() const = 0;
    friend std::ostream &operator<<(std::ostream &os, const NDTFeatureLinkInterface &obj)
    {
        os << "\n[" << obj.getRefIdx() << "<->" << obj.getMovIdx() << "] T: " << "[" << obj.getRelPose().translation()[0] << ","
           << obj.getRelPose().translation()[1] << "](" << obj.getRelPose().rotation().eulerAngles(0, 1, 2)[2] << ")";
        os << "\n score : " << obj.getScore();
        return os;
    }
};
class NDTFeatureGraphInterface {
public:
    virtual ~NDTFeatureGraphInterface() {}
    virtual size_t getNbNodes() const = 0;
    virtual NDTFeatureNodeInterface &getNodeInterface(size_t idx) = 0;
    virtual const NDTFeatureNodeInterface &getNodeInterface(size_t idx) const = 0;
    virtual size_t getNbLinks() const = 0;
    virtual NDTFeatureLinkInterface &getLinkInterface(size_t idx) = 0;
    virtual const NDTFeatureLinkInterface &getLinkInterface(size_t idx) const = 0;
};

class NDTFeatureLink : public NDTFeatureLinkInterface {
public:
    NDTFeatureLink() : ref_idx(0), mov_idx(0), cov_3d(6, 6), score(0.) {}
    NDTFeatureLink(size_t ref, size_t mov) : ref_idx(ref), mov_idx(mov), cov_3d(6, 6), score(0.) {}
    size_t ref_idx, mov_idx;   // Vector idx...
    Eigen::Affine3d T;         // From ref->mov.
    Eigen::Matrix3d cov;
    Eigen::MatrixXd cov_3d;    // the covariance returned by NDTMatcherD2D::covariance (graph.cpp:296-310, 330)
    double score;
    bool converged = true;     // (added) replaces the stdin pause of graph.cpp:318-323
    int iterations = 0;
    // Interfaces.
    virtual const Eigen::Affine3d &getRelPose() const { return T; }
    virtual const Eigen::Matrix3d &getRelCov() const { return cov; }
    virtual double getScore() const { return score; }
    virtual size_t getRefIdx() const { return ref_idx; }
    virtual size_t getMovIdx() const { return mov_idx; }
    void force2D() { forceEigenAffine3dTo2dInPlace(this->T); }
};

class NDTFeatureNode : public NDTFeatureNodeInterface {
public:
    NDTFeatureNode() : map(NULL), nbUpdates(0), time_last_update(0) {}
    NDTFeatureFuserHMT *map;   // owned by the graph (ndt_feature_graph.h:78-88); node copies are shallow like the reference's
    Eigen::Affine3d T;
    Eigen::Matrix3d cov;
    Eigen::Affine3d Tlocal_odom;   // Incremental odometry between successive local maps.
    Eigen::Affine3d Tlocal_fuse;   // Incremental fuse estimates between sucessive local maps.
    pcl::PointCloud<pcl::PointXYZ> pts;   // Only for visualizaion purposes...
    int nbUpdates;
    double time_last_update;
    // ndt_feature_node.h:81-96
    void addCloud(Eigen::Affine3d &T_, const pcl::PointCloud<pcl::PointXYZ> &pc)
    {
        pcl::PointCloud<pcl::PointXYZ> moved(pc);
        ndtgpu_host::transformPointCloudInPlace(T_, moved);
        for (const auto &p : moved.points) this->pts.push_back(p);
    }
    pcl::PointCloud<pcl::PointXYZ> getGlobalPointCloud()
    {
        pcl::PointCloud<pcl::PointXYZ> out(this->pts);
        ndtgpu_host::transformPointCloudInPlace(this->T, out);
        return out;
    }
    pcl::PointCloud<pcl::PointXYZ> getLocalPointCloud() { return this->pts; }
    lslgeneric::NDTMap &getNDTMap() { return *(map->map); }
    NDTFeatureFuserHMT &getFuser() { return *map; }
    // Interfaces.
    virtual const Eigen::Affine3d &getPose() const { return T; }
    virtual void setPose(const Eigen::Affine3d &pose) { T = pose; }
    virtual const Eigen::Matrix3d &getCov() const { return cov; }
    virtual void setCov(const Eigen::Matrix3d &c) { cov = c; }
    void force2D()
    {
        forceEigenAffine3dTo2dInPlace(this->T);
        forceEigenAffine3dTo2dInPlace(this->Tlocal_odom);
        forceEigenAffine3dTo2dInPlace(this->Tlocal_fuse);
    }
};

// ndt_feature::overlapNDTOccupancyScore(ref, mov, T)  (ndt_feature_node.h:213-252), on the device
inline double overlapNDTOccupancyScore(NDTFeatureNode &ref, NDTFeatureNode &mov, const Eigen::Affine3d &T)
{
    uint32_t ri = (uint32_t)ref.getNDTMap().slot(), mi = (uint32_t)mov.getNDTMap().slot();
    double score = 1.;
    ndtgpu_host::check(ndtgpu_overlap_score_batch(ref.getNDTMap().handle(), &ri, mov.getNDTMap().handle(), &mi, T.data(), 1, &score, nullptr, nullptr),
                       "ndtgpu_overlap_score_batch");
    return score;
}

class NDTFeatureGraph : public NDTFeatureGraphInterface {
public:
    class Params {   // ndt_feature_graph.h:24-56
    public:
        Params()
        {
            newNodeNumberOfFrames = 20;
            newNodeTranslDist = 1.;
            storePtsInNodes = false;
            storePtsInNodesIncr = 8;
            popNodes = false;
            maxNodes = 256;
        }
        int newNodeNumberOfFrames;
        double newNodeTranslDist;
        bool storePtsInNodes;
        int storePtsInNodesIncr;
        bool popNodes;
        size_t maxNodes;   // (added) capacity of the device pool the node maps live in
    };

    NDTFeatureGraph() : distance_moved_in_last_node_(0.) {}
    NDTFeatureGraph(const NDTFeatureGraph::Params &params, const NDTFeatureFuserHMT::Params &fuserParams)
        : params_(params), fuser_params_(fuserParams), distance_moved_in_last_node_(0.) {}
    virtual ~NDTFeatureGraph()
    {
        for (auto it = nodes_.begin(); it != nodes_.end(); ++it)
            if (it->map != NULL) delete it->map;
    }
    NDTFeatureGraph(const NDTFeatureGraph &) = delete;
    NDTFeatureGraph &operator=(const NDTFeatureGraph &) = delete;

    // Interfaces (ndt_feature_graph.h: the graph IS-A NDTFeatureGraphInterface)
    virtual size_t getNbNodes() const { return nodes_.size(); }
    virtual NDTFeatureNodeInterface &getNodeInterface(size_t idx) { return nodes_[idx]; }
    virtual const NDTFeatureNodeInterface &getNodeInterface(size_t idx) const { return nodes_[idx]; }
    virtual size_t getNbLinks() const { return links_.size(); }
    virtual NDTFeatureLinkInterface &getLinkInterface(size_t idx) { return links_[idx]; }
    virtual const NDTFeatureLinkInterface &getLinkInterface(size_t idx) const { return links_[idx]; }
    NDTFeatureNode &getNode(size_t i) { return nodes_[i]; }
    NDTFeatureLink &getLink(size_t i) { return links_[i]; }
    lslgeneric::NDTMap *getMap(int i) { return nodes_[i].map->map; }
    Eigen::Affine3d getT() { return Tnow; }
    void setFuserParams(const NDTFeatureFuserHMT::Params &p) { fuser_params_ = p; }
    void setMotionParams(const MotionModel2d::Params &p) { motion_params_ = p; }
    void setSensorPose(const Eigen::Affine3d &p) { sensor_pose_ = p; }
    void clearAllLinks() { links_.clear(); }
    void setLinks(const std::vector<NDTFeatureLink> &l) { links_ = l; }
    void appendLinks(const std::vector<NDTFeatureLink> &l) { links_.insert(links_.end(), l.begin(), l.end()); }
    const std::vector<NDTFeatureLink> &getCurrentLinks() { return links_; }
    void force2D()
    {
        for (auto &n : nodes_) n.force2D();
        for (auto &l : links_) l.force2D();
    }
    bool fullInit()
    {
        for (size_t i = 0; i < getNbNodes(); ++i)
            if (nodes_[i].map->wasInit() == false) return false;
        return true;
    }

    // initialize(initPose, cloud, pts, preLoad) -- ndt_feature_graph.cpp:24-55: the first node.  A node keeps its map in its
    // own frame (node.T places it in the world), so its fuser starts at the identity.
    void initialize(Eigen::Affine3d initPose, pcl::PointCloud<pcl::PointXYZ> &cloud, const InterestPointVec &pts, bool preLoad = false)
    {
        open_node(initPose, cloud, pts, preLoad);
        Tnow = initPose;
    }

    // update(Tmotion, cloud, pts) -- ndt_feature_graph.cpp:60-144: the scan goes to the current node's fuser; once the robot
    // has moved newNodeTranslDist inside a node, the scan is only LOCALISED there (no fuse-in) and becomes the first scan of
    // a new node at the pose that localisation gave.  Returns the pose in world coordinates.
    Eigen::Affine3d update(Eigen::Affine3d Tmotion, pcl::PointCloud<pcl::PointXYZ> &cloud, const InterestPointVec &pts)
    {
        distance_moved_in_last_node_ += Tmotion.translation().norm();
        const bool leave = distance_moved_in_last_node_ > params_.newNodeTranslDist;
        NDTFeatureNode &cur = nodes_.back();
        const Eigen::Affine3d local = cur.map->update(Tmotion, cloud, pts, !leave, !leave);
        Tnow = cur.T * local;
        cur.Tlocal_odom = cur.Tlocal_odom * Tmotion;
        cur.Tlocal_fuse = local;
        if (leave) {
            distance_moved_in_last_node_ = 0.;
            if (params_.popNodes) {          // (only the newest node is kept)
                delete nodes_.back().map;
                nodes_.pop_back();
                pool_->release_last();
            }
            open_node(Tnow, cloud, pts, false);
            return Tnow;
        }
        if (params_.storePtsInNodes && cur.nbUpdates % params_.storePtsInNodesIncr == 0) {
            Eigen::Affine3d in_node = local * sensor_pose_;
            cur.addCloud(in_node, cloud);
        }
        cur.nbUpdates++;
        return Tnow;
    }

    // computeLink (graph.cpp:162-177) without the FLIRT feature match that seeds link.T upstream (out of scope): the
    // relative pose of the node estimates, scored by the occupancy overlap like the reference
    NDTFeatureLink computeLink(size_t idx_ref, size_t idx_mov)
    {
        NDTFeatureLink m(idx_ref, idx_mov);
        m.T = nodes_[idx_ref].T.inverse() * nodes_[idx_mov].T;
        m.score = overlapNDTOccupancyScore(nodes_[idx_ref], nodes_[idx_mov], m.T);
        return m;
    }
    // computeAllPossibleLinks (graph.cpp:395-405); the overlap scores of all links in ONE device call
    std::vector<NDTFeatureLink> computeAllPossibleLinks()
    {
        std::vector<NDTFeatureLink> ret;
        for (size_t i = 0; i < nodes_.size(); i++)
            for (size_t j = i + 1; j < nodes_.size(); j++) {
                NDTFeatureLink m(i, j);
                m.T = nodes_[i].T.inverse() * nodes_[j].T;
                ret.push_back(m);
            }
        score_links(ret);
        return ret;
    }
    std::vector<NDTFeatureLink> getIncrementalLinks() const   // graph.cpp:180-204
    {
        std::vector<NDTFeatureLink> ret;
        for (size_t i = 0; i + 1 < nodes_.size(); i++) {
            NDTFeatureLink m(i, i + 1);
            m.T = nodes_[i].T.inverse() * nodes_[i + 1].T;
            m.score = -1.;
            ret.push_back(m);
        }
        return ret;
    }

    // updateLinkUsingNDTRegistration (graph.cpp:260-345): one link
    void updateLinkUsingNDTRegistration(NDTFeatureLink &link, int nb_neighbours, bool keepScore)
    {
        std::vector<NDTFeatureLink> one(1, link);
        updateLinksUsingNDTRegistration(one, nb_neighbours, keepScore);
        link = one[0];
    }
    // updateLinksUsingNDTRegistration (graph.cpp:347-353): every link's match, covariance and overlap score in three
    // batched device calls
    void updateLinksUsingNDTRegistration(std::vector<NDTFeatureLink> &links, int nb_neighbours, bool keepScore)
    {
        const size_t n = links.size();
        if (!n) return;
        std::vector<uint32_t> ti(n), si(n);
        std::vector<double> T(16 * n), before(16 * n), cov(36 * n);
        std::vector<ndtgpu_match_result> res(n);
        std::vector<int32_t> singular(n);
        for (size_t k = 0; k < n; k++) {
            ti[k] = (uint32_t)nodes_[links[k].ref_idx].getNDTMap().slot();
            si[k] = (uint32_t)nodes_[links[k].mov_idx].getNDTMap().slot();
            const double *m = links[k].T.data();
            for (int q = 0; q < 16; q++) T[16 * k + q] = before[16 * k + q] = m[q];
        }
        lslgeneric::NDTMatcherD2D matcher_d2d;          // default-constructed, graph.cpp:261
        matcher_d2d.n_neighbours = nb_neighbours;       // graph.cpp:262
        ndtgpu_match_params p = matcher_d2d.params(0x3f, true);
        ndtgpu_mapset *pool = pool_->handle();
        ndtgpu_host::check(ndtgpu_match_batch(pool, ti.data(), pool, si.data(), T.data(), n, &p, res.data(), nullptr), "ndtgpu_match_batch");
        ndtgpu_host::check(ndtgpu_covariance_batch(pool, ti.data(), pool, si.data(), T.data(), n, &p, matcher_d2d.covariance_mode, cov.data(),
                                                   singular.data(), nullptr), "ndtgpu_covariance_batch");
        for (size_t k = 0; k < n; k++) {
            double *m = links[k].T.data();
            bool same = true;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) same = same && (before[16 * k + c * 4 + r] == T[16 * k + c * 4 + r]);
            for (int q = 0; q < 16; q++) m[q] = T[16 * k + q];
            links[k].converged = res[k].converged != 0;
            links[k].iterations = res[k].iterations;
            if (links[k].cov_3d.rows() != 6) links[k].cov_3d.resize(6, 6);
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++)   // "NOTHING HAPPENED": 0.02 * identity, graph.cpp:300-310
                    links[k].cov_3d(a, b) = same ? ((a == b) ? 0.02 : 0.0) : cov[36 * k + a * 6 + b];
        }
        if (!keepScore) score_links(links);             // graph.cpp:335-341
    }
    void updateAllGraphLinksUsingNDTRegistration(int nb_neighbours, bool keepScore)
    {
        // (the reference updates COPIES of its links here, graph.cpp:249-258: the results are lost; the stored links are
        //  updated instead)
        updateLinksUsingNDTRegistration(links_, nb_neighbours, keepScore);
    }

    // getValidLinks (graph.cpp:527-556)
    std::vector<NDTFeatureLink> getValidLinks(const std::vector<NDTFeatureLink> &links, double maxScore, double maxDist,
                                              double maxAngularDist, int minIdxDist) const
    {
        std::vector<NDTFeatureLink> ret;
        for (size_t i = 0; i < links.size(); i++) {
            if (links[i].getScore() > maxScore) continue;
            if (std::abs((int)links[i].getMovIdx() - (int)links[i].getRefIdx()) < minIdxDist) continue;
            const NDTFeatureNode &ref_node = nodes_[links[i].getRefIdx()];
            const NDTFeatureNode &mov_node = nodes_[links[i].getMovIdx()];
            Eigen::Affine3d Tlink = ref_node.T * links[i].T;
            double dist, angular_dist;
            distanceBetweenAffine3d(mov_node.T, Tlink, dist, angular_dist);
            if (dist < maxDist && angular_dist < maxAngularDist) ret.push_back(links[i]);
        }
        return ret;
    }

    Params params_;
    NDTFeatureFuserHMT::Params fuser_params_;
    MotionModel2d::Params motion_params_;

protected:
    // the node fusers are the slots of ONE bank over the pool the node maps live in
    NDTFeatureFuserHMT *new_fuser()
    {
        if (!pool_) {
            const double c[3] = {0, 0, 0}, s[3] = {fuser_params_.map_size_x, fuser_params_.map_size_y, fuser_params_.map_size_z};
            pool_ = std::make_shared<ndtgpu_host::MapPool>(fuser_params_.resolution, c, s, params_.maxNodes);
            bank_ = std::make_shared<ndtgpu_host::FuserBank>(NDTFeatureFuserHMT::bankParams(fuser_params_, motion_params_, sensor_pose_), pool_);
        }
        return new NDTFeatureFuserHMT(fuser_params_, bank_, pool_->allocate());
    }
    // a new node at world pose `at`, its map started from `cloud`
    void open_node(const Eigen::Affine3d &at, pcl::PointCloud<pcl::PointXYZ> &cloud, const InterestPointVec &pts, bool preLoad)
    {
        NDTFeatureNode node;
        node.T = at;
        node.map = new_fuser();
        node.map->setMotionParams(motion_params_);
        node.map->setSensorPose(sensor_pose_);
        node.map->initialize(Eigen::Affine3d::Identity(), cloud, pts, preLoad);
        if (params_.storePtsInNodes && nodes_.empty()) {
            Eigen::Affine3d in_node = sensor_pose_;
            node.addCloud(in_node, cloud);
        }
        nodes_.push_back(node);
    }
    void score_links(std::vector<NDTFeatureLink> &links)
    {
        const size_t n = links.size();
        if (!n) return;
        std::vector<uint32_t> ri(n), mi(n);
        std::vector<double> T(16 * n), score(n);
        for (size_t k = 0; k < n; k++) {
            ri[k] = (uint32_t)nodes_[links[k].ref_idx].getNDTMap().slot();
            mi[k] = (uint32_t)nodes_[links[k].mov_idx].getNDTMap().slot();
            const double *m = links[k].T.data();
            for (int q = 0; q < 16; q++) T[16 * k + q] = m[q];
        }
        ndtgpu_host::check(ndtgpu_overlap_score_batch(pool_->handle(), ri.data(), pool_->handle(), mi.data(), T.data(), n, score.data(), nullptr, nullptr),
                           "ndtgpu_overlap_score_batch");
        for (size_t k = 0; k < n; k++) links[k].score = score[k];
    }

    std::shared_ptr<ndtgpu_host::MapPool> pool_;
    std::shared_ptr<ndtgpu_host::FuserBank> bank_;
    std::vector<NDTFeatureNode> nodes_;
    std::vector<NDTFeatureLink> links_;
    Eigen::Affine3d sensor_pose_, Tnow;
    double distance_moved_in_last_node_;
};

}  // namespace ndt_feature
