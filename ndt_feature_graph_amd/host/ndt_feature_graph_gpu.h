// ndt_feature_graph_gpu.h -- host mirror of the graph-layer entry points that drive the hot path:
// ndt_feature::NDTFeatureLink / NDTFeatureNode / NDTFeatureGraph (ndt_feature_link.h:9-56,
// ndt_feature_node.h:38-209, ndt_feature_graph.h:20-280), restricted to what the NDT path needs:
// node maps, candidate links, the registration of links (ndt_feature_graph.cpp:260-353) and the
// link gates (:527-556).  The ROS / iSAM / FLIRT members of the reference classes stay where they
// are (out of scope); the incremental fuser update is a "next" row (8f-1).
//
// What changes against the reference: updateLinksUsingNDTRegistration registers ALL links in one
// batched GPU call instead of a serial loop, never blocks on stdin (graph.cpp:318-328) and reports
// non-convergence in NDTFeatureLink::converged.
#pragma once
#include "lslgeneric_gpu.h"

#include <cstdlib>

namespace ndt_feature {

using ndtgpu_host::Affine3d;
using ndtgpu_host::PointCloud;
using ndtgpu_host::PointXYZ;

// getRobustYawFromAffine3d (utils.h:30-40) for rigid transforms
inline double getRobustYawFromAffine3d(const Affine3d &a) { return std::atan2(a(1, 0), a(0, 0)); }
// distanceBetweenAffine3d (utils.h:42-47)
inline void distanceBetweenAffine3d(const Affine3d &p1, const Affine3d &p2, double &dist, double &angularDist)
{
    Affine3d tmp = p1.inverse() * p2;
    dist = std::sqrt(tmp(0, 3) * tmp(0, 3) + tmp(1, 3) * tmp(1, 3) + tmp(2, 3) * tmp(2, 3));
    angularDist = std::fabs(getRobustYawFromAffine3d(tmp));
}

// ndt_feature::matchFusion (ndt_matcher_d2d_fusion.h:797-1155) for the shipped configurations
// (useNDT, no FLIRT features, no Tikhonov): D2D-NDT + the odometry soft constraint.  The feature-map
// arguments of the reference signature are dropped (they are empty maps there).
inline bool matchFusion(lslgeneric::NDTMap &targetNDT, lslgeneric::NDTMap &sourceNDT, Affine3d &T, const double Tcov[36],
                        bool useInitialGuess, bool step_control, int ITR_MAX = 30, int n_neighbours = 2,
                        double DELTA_SCORE = 10e-4, bool useSoftConstraints = true)
{
    lslgeneric::NDTMatcherD2D m;
    m.n_neighbours = n_neighbours;
    m.ITR_MAX = ITR_MAX;
    m.DELTA_SCORE = DELTA_SCORE;
    m.step_control = step_control;
    ndtgpu_match_params p = m.params(0x3f, useInitialGuess);
    ndtgpu_match_result r;
    uint32_t ti = (uint32_t)targetNDT.slot(), si = (uint32_t)sourceNDT.slot();
    ndtgpu_host::check(ndtgpu_match_fusion_batch(targetNDT.handle(), &ti, sourceNDT.handle(), &si, ndtgpu_host::affine_data(T),
                                                 Tcov, 1, &p, useSoftConstraints ? 1 : 0, &r, nullptr), "ndtgpu_match_fusion_batch");
    return r.converged != 0;
}

class NDTFeatureLink {
public:
    NDTFeatureLink() : ref_idx(0), mov_idx(0), score(0.) {}
    NDTFeatureLink(size_t ref, size_t mov) : ref_idx(ref), mov_idx(mov), score(0.) {}
    size_t ref_idx, mov_idx;
    Affine3d T;                 // from ref -> mov
    double cov_3d[36] = {0};    // NDTMatcherD2D::covariance is a "next" row; 0.02*I fallback of graph.cpp:300-310
    double score;
    bool converged = true;      // replaces the stdin pause of graph.cpp:318-323
    int iterations = 0;
    const Affine3d &getRelPose() const { return T; }
    double getScore() const { return score; }
    size_t getRefIdx() const { return ref_idx; }
    size_t getMovIdx() const { return mov_idx; }
};

class NDTFeatureNode {
public:
    std::shared_ptr<lslgeneric::NDTMap> map;   // reference: NDTFeatureFuserHMT* map; map->map is the NDTMap
    Affine3d T, Tlocal_odom, Tlocal_fuse;
    int nbUpdates = 0;
    lslgeneric::NDTMap &getNDTMap() { return *map; }
    const Affine3d &getPose() const { return T; }
};

class NDTFeatureGraph {
public:
    struct Params {
        double newNodeTranslDist = 1.;   // ndt_feature_graph.h:28
        double resolution = 1.;          // fuser params (ndt_feature_fuser_hmt.h:63-67)
        double map_size_x = 40., map_size_y = 40., map_size_z = 10.;
        double sensor_range = 3.;
        size_t max_nodes = 64;
        uint32_t max_cells = 0;
    };

    explicit NDTFeatureGraph(const Params &p) : params_(p)
    {
        double c[3] = {0, 0, 0}, s[3] = {p.map_size_x, p.map_size_y, p.map_size_z};
        pool_ = std::make_shared<ndtgpu_host::MapPool>(p.resolution, c, s, p.max_nodes, p.max_cells);
    }

    size_t getNbNodes() const { return nodes_.size(); }
    size_t getNbLinks() const { return links_.size(); }
    NDTFeatureNode &getNode(size_t i) { return nodes_[i]; }
    NDTFeatureLink &getLink(size_t i) { return links_[i]; }
    lslgeneric::NDTMap *getMap(int i) { return nodes_[i].map.get(); }
    Affine3d getT() { return nodes_.back().T; }
    void clearAllLinks() { links_.clear(); }
    void setLinks(const std::vector<NDTFeatureLink> &l) { links_ = l; }
    void appendLinks(const std::vector<NDTFeatureLink> &l) { links_.insert(links_.end(), l.begin(), l.end()); }
    const std::vector<NDTFeatureLink> &getCurrentLinks() { return links_; }

    // New node whose map is built from one cloud in the node frame (the per-scan local map of
    // NDTFeatureFuserHMT::update, fuser_hmt.cpp:195-227: guessSize + loadPointCloud + computeNDTCells).
    size_t addNode(const Affine3d &pose, const PointCloud<PointXYZ> &cloud)
    {
        NDTFeatureNode n;
        n.T = pose;
        n.map = std::make_shared<lslgeneric::NDTMap>(pool_, pool_->allocate());
        n.map->guessSize(0, 0, 0, params_.map_size_x, params_.map_size_y, params_.map_size_z);
        n.map->loadPointCloud(cloud, params_.sensor_range);
        n.map->computeNDTCells(lslgeneric::CELL_UPDATE_MODE_SAMPLE_VARIANCE);
        n.nbUpdates = 1;
        nodes_.push_back(n);
        return nodes_.size() - 1;
    }

    // computeAllPossibleLinks (graph.cpp:395-405) with the odometry-predicted relative pose as link.T
    // (the reference seeds it from the FLIRT feature match, computeLink :162-177 -- out of scope)
    std::vector<NDTFeatureLink> computeAllPossibleLinks()
    {
        std::vector<NDTFeatureLink> ret;
        for (size_t i = 0; i < nodes_.size(); i++)
            for (size_t j = i + 1; j < nodes_.size(); j++) {
                NDTFeatureLink m(i, j);
                m.T = nodes_[i].T.inverse() * nodes_[j].T;
                m.score = -1.;
                ret.push_back(m);
            }
        return ret;
    }

    // updateLinkUsingNDTRegistration (graph.cpp:260-345): one link
    void updateLinkUsingNDTRegistration(NDTFeatureLink &link, int nb_neighbours, bool /*keepScore*/)
    {
        std::vector<NDTFeatureLink> one(1, link);
        updateLinksUsingNDTRegistration(one, nb_neighbours, true);
        link = one[0];
    }

    // updateLinksUsingNDTRegistration (graph.cpp:347-353): every link in ONE batched GPU call
    void updateLinksUsingNDTRegistration(std::vector<NDTFeatureLink> &links, int nb_neighbours, bool /*keepScore*/)
    {
        const size_t n = links.size();
        if (!n) return;
        std::vector<uint32_t> ti(n), si(n);
        std::vector<double> T(16 * n);
        std::vector<ndtgpu_match_result> res(n);
        for (size_t k = 0; k < n; k++) {
            ti[k] = (uint32_t)nodes_[links[k].ref_idx].map->slot();
            si[k] = (uint32_t)nodes_[links[k].mov_idx].map->slot();
            const double *m = ndtgpu_host::affine_data(links[k].T);
            for (int q = 0; q < 16; q++) T[16 * k + q] = m[q];
        }
        lslgeneric::NDTMatcherD2D matcher_d2d;          // default-constructed, graph.cpp:261
        matcher_d2d.n_neighbours = nb_neighbours;       // graph.cpp:262
        ndtgpu_match_params p = matcher_d2d.params(0x3f, true);
        ndtgpu_host::check(ndtgpu_match_batch(pool_->handle(), ti.data(), pool_->handle(), si.data(), T.data(), n, &p,
                                              res.data(), nullptr), "ndtgpu_match_batch");
        for (size_t k = 0; k < n; k++) {
            double *m = ndtgpu_host::affine_data(links[k].T);
            bool same = true;
            for (int q = 0; q < 16; q++) { same = same && (m[q] == T[16 * k + q]); m[q] = T[16 * k + q]; }
            links[k].converged = res[k].converged != 0;
            links[k].iterations = res[k].iterations;
            if (same)   // "NOTHING HAPPENED": identity-scaled covariance, graph.cpp:300-310
                for (int q = 0; q < 36; q++) links[k].cov_3d[q] = (q % 7 == 0) ? 0.02 : 0.0;
        }
    }
    void updateAllGraphLinksUsingNDTRegistration(int nb_neighbours, bool keepScore)
    {
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
            Affine3d Tlink = ref_node.T * links[i].T;
            double dist, angular_dist;
            distanceBetweenAffine3d(mov_node.T, Tlink, dist, angular_dist);
            if (dist < maxDist && angular_dist < maxAngularDist) ret.push_back(links[i]);
        }
        return ret;
    }

    Params params_;

protected:
    std::shared_ptr<ndtgpu_host::MapPool> pool_;
    std::vector<NDTFeatureNode> nodes_;
    std::vector<NDTFeatureLink> links_;
};

}  // namespace ndt_feature
