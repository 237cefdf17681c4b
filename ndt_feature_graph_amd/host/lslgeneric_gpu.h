// lslgeneric_gpu.h -- host C++ mirror of the lslgeneric:: classes that MalcolmMielle/ndt_feature_graph
// calls on its hot path, implemented over the C-ABI of libndtgpu.so (include/ndtgpu.h).
//
// Same class and member names, argument types and meaning, ownership and error behaviour as the call sites in
//   ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:87-94, 195-227, 403-405, 485-486
//   ndt_feature/src/ndt_feature_src/ndt_feature_graph.cpp:261-298
//   ndt_feature/include/ndt_feature/ndt_matcher_d2d_fusion.h:811-814, 840-841, 856, 953-962, 1013, 1047-1056, 1085, 1170-1175
//   ndt_feature/src/ndt_odom_debug.cpp:163-206
// so that a maintainer can point those translation units at this header instead of
// <ndt_map/ndt_map.h>, <ndt_map/lazy_grid.h>, <ndt_registration/ndt_matcher_d2d.h>, <ndt_registration/ndt_matcher_d2d_2d.h>.
// host/host_demo.cpp (part C) re-types the Newton loop of ndt_matcher_d2d_fusion.h:847-1121 against it.
// Everything that computes lives on the GPU; there is no CPU fallback: a failed C-ABI call throws
// ndtgpu_host::Error (the reference has no error channel here besides bool returns).  Arguments the reference passes
// but this implementation cannot honour are REJECTED (Error, NDTGPU_ERR_INVALID), never ignored.
#pragma once
#include "../../include/ndtgpu.h"
#include "ndt_gpu_types.h"

#include <array>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ndtgpu_host {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};
inline void check(ndtgpu_status s, const char *where)
{
    if (s != NDTGPU_OK) throw Error(s, std::string(where) + ": " + ndtgpu_last_error());
}

// B maps with one geometry in one device arena (a node-map pool); NDTMap objects index into it.
class MapPool {
public:
    MapPool(double res, const double centre[3], const double size_m[3], size_t n_maps, uint32_t max_cells = 0)
        : res_(res), n_maps_(n_maps)
    {
        ndtgpu_grid_params g;
        g.res = res;
        for (int a = 0; a < 3; a++) { g.centre[a] = centre[a]; g.size[a] = size_m[a]; size_[a] = size_m[a]; }
        g.max_cells = max_cells;
        check(ndtgpu_mapset_create(&g, n_maps, &set_), "ndtgpu_mapset_create");
    }
    ~MapPool() { ndtgpu_mapset_destroy(set_); }
    MapPool(const MapPool &) = delete;
    MapPool &operator=(const MapPool &) = delete;
    ndtgpu_mapset *handle() const { return set_; }
    size_t size() const { return n_maps_; }
    double resolution() const { return res_; }
    const double *size_m() const { return size_; }
    size_t allocate()
    {
        if (next_ >= n_maps_) throw Error(NDTGPU_ERR_CAPACITY, "MapPool exhausted");
        return next_++;
    }
    void release_last() { if (next_) next_--; }      // NDTFeatureGraph::Params::popNodes

private:
    ndtgpu_mapset *set_ = nullptr;
    double res_, size_[3];
    size_t n_maps_, next_ = 0;
};

}  // namespace ndtgpu_host

namespace lslgeneric {

enum NDTCellUpdateMode { CELL_UPDATE_MODE_COVARIANCE_INTERSECTION, CELL_UPDATE_MODE_SAMPLE_VARIANCE };

// one Gaussian on the host (NDTCell::getMean / getCov / setMean / setCov / hasGaussian_)
class NDTCell {
public:
    bool hasGaussian_ = true;
    NDTCell() {}
    Eigen::Vector3d getMean() const { return mean_; }
    Eigen::Matrix3d getCov() const { return cov_; }
    void setMean(const Eigen::Vector3d &m) { mean_ = m; }
    void setCov(const Eigen::Matrix3d &c) { cov_ = c; }
    int getN() const { return n_; }
    void setN(int n) { n_ = n; }
    int idx[3] = {0, 0, 0};   // LazyGrid cell index (cells read from a map)

private:
    Eigen::Vector3d mean_;
    Eigen::Matrix3d cov_;
    int n_ = 0;
};

// LazyGrid(res): only carries the cell size (the dense table lives on the device)
class LazyGrid {
public:
    explicit LazyGrid(double cellSize) : res(cellSize) {}
    double res;
};
using SpatialIndex = LazyGrid;

// lslgeneric::CellVector: a list of Gaussians without a grid (the feature / odometry-cell maps of matchFusion,
// ndt_feature_fuser_hmt.cpp:291-334).  Lives on the host: these maps hold a few dozen cells with known correspondence.
class CellVector {
public:
    CellVector() {}
    ~CellVector() { for (NDTCell *c : cells_) delete c; }
    CellVector(const CellVector &) = delete;
    CellVector &operator=(const CellVector &) = delete;
    void addCell(NDTCell *cell) { cells_.push_back(cell); }          // takes ownership
    void addNDTCell(NDTCell *cell) { cells_.push_back(cell); }
    NDTCell *getCellIdx(unsigned int idx) const { return idx < cells_.size() ? cells_[idx] : NULL; }
    int size() const { return (int)cells_.size(); }

private:
    std::vector<NDTCell *> cells_;
};

class NDTMap {
public:
    // new NDTMap(new LazyGrid(res))  -- takes ownership of idx like the reference (fuser_hmt.cpp:87, 195-196)
    explicit NDTMap(SpatialIndex *idx, bool /*dealloc*/ = false) : res_(idx->res) { delete idx; }
    // NDTMap(CellVector*, dealloc) (fuser_hmt.cpp:303-304): a map over a cell list; only the cell-list calls below apply
    explicit NDTMap(CellVector *cv, bool dealloc = false) : res_(0.), cv_(cv), cv_owned_(dealloc) {}
    ~NDTMap() { if (cv_owned_) delete cv_; }
    NDTMap(const NDTMap &) = delete;
    NDTMap &operator=(const NDTMap &) = delete;
    CellVector *getMyIndex() const { return cv_; }
    // NDTMap::pseudoTransformNDTMap(T) of a cell-list map: a new map (caller deletes) with mean' = T mean, cov' = R cov R^T
    NDTMap *pseudoTransformNDTMap(const Eigen::Affine3d &T) const
    {
        if (!cv_) throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTMap::pseudoTransformNDTMap: only for CellVector maps (grid maps: pseudoTransformNDT)");
        CellVector *out = new CellVector();
        const double *m = T.data();
        for (int i = 0; i < cv_->size(); i++) {
            const NDTCell *c = cv_->getCellIdx((unsigned)i);
            Eigen::Vector3d mu = c->getMean(), t;
            Eigen::Matrix3d C = c->getCov(), R, RC, RCRt;
            for (int r = 0; r < 3; r++) {
                t(r) = m[0 * 4 + r] * mu(0) + m[1 * 4 + r] * mu(1) + m[2 * 4 + r] * mu(2) + m[12 + r];
                for (int k = 0; k < 3; k++) R(r, k) = m[k * 4 + r];
            }
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) RC(r, k) = R(r, 0) * C(0, k) + R(r, 1) * C(1, k) + R(r, 2) * C(2, k);
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) RCRt(r, k) = RC(r, 0) * R(k, 0) + RC(r, 1) * R(k, 1) + RC(r, 2) * R(k, 2);
            NDTCell *n = new NDTCell();
            n->setMean(t); n->setCov(RCRt); n->setN(c->getN());
            out->addCell(n);
        }
        return new NDTMap(out, true);
    }
    // a map that lives in a shared pool (graph node maps, the fuser's per-scan map): geometry comes from the pool
    NDTMap(std::shared_ptr<ndtgpu_host::MapPool> pool, size_t slot) : pool_(std::move(pool)), slot_(slot), res_(pool_->resolution())
    {
        for (int a = 0; a < 3; a++) size_[a] = pool_->size_m()[a];
        have_size_ = true;
    }

    // NDTMap::initialize(cx,cy,cz,sx,sy,sz)  (fuser_hmt.cpp:89): every cell exists from now on and carries an
    // occupancy; addPointCloud ray-traces (isFirstLoad_ == false)
    void initialize(double cx, double cy, double cz, double sx, double sy, double sz)
    {
        set_geometry(cx, cy, cz, sx, sy, sz);
        ndtgpu_host::check(ndtgpu_mapset_enable_occupancy(handle()), "ndtgpu_mapset_enable_occupancy");
        ndtgpu_host::check(ndtgpu_mapset_clear(handle(), slot_, 1), "ndtgpu_mapset_clear");
        initialized_ = true;
    }
    // NDTMap::guessSize(cx,cy,cz,sx,sy,sz)  (fuser_hmt.cpp:222): explicit centre + extent for loadPointCloud
    void guessSize(double cx, double cy, double cz, double sx, double sy, double sz) { set_geometry(cx, cy, cz, sx, sy, sz); }
    // NDTMap::setMapSize (ndt_odom_debug.cpp:177): extent only, centre = centroid of the cloud
    void setMapSize(double sx, double sy, double sz)
    {
        if (pool_ && (sx != size_[0] || sy != size_[1] || sz != size_[2])) throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTMap::setMapSize: a pooled map has the pool's extent");
        size_[0] = sx; size_[1] = sy; size_[2] = sz; have_size_ = true;
    }

    // NDTMap::loadPointCloud(cloud, range_limit)  (fuser_hmt.cpp:225; ndt_odom_debug.cpp:178).  The points are copied
    // (like NDTCell::points_); the cells appear at computeNDTCells.
    void loadPointCloud(const pcl::PointCloud<pcl::PointXYZ> &pc, double range_limit = -1.)
    {
        if (!have_geometry_) {   // guess_size_ == true upstream: centre = centroid of the accepted points
            double c[3] = {0, 0, 0};
            size_t n = 0;
            double maxd = 0;
            for (const auto &p : pc.points) {
                if (std::isnan(p.x) || std::isnan(p.y) || std::isnan(p.z)) continue;
                if (range_limit > 0 && std::sqrt((double)p.x * p.x + (double)p.y * p.y + (double)p.z * p.z) > range_limit) continue;
                c[0] += p.x; c[1] += p.y; c[2] += p.z; n++;
            }
            if (n) for (int a = 0; a < 3; a++) c[a] /= (double)n;
            for (const auto &p : pc.points) {
                if (std::isnan(p.x) || std::isnan(p.y) || std::isnan(p.z)) continue;
                double d = std::sqrt((c[0] - p.x) * (c[0] - p.x) + (c[1] - p.y) * (c[1] - p.y) + (c[2] - p.z) * (c[2] - p.z));
                if (d > maxd) maxd = d;
            }
            double s[3] = {size_[0], size_[1], size_[2]};
            if (!have_size_) s[0] = s[1] = s[2] = 4 * maxd;
            set_geometry(c[0], c[1], c[2], s[0], s[1], s[2]);
            have_geometry_ = false;                // a later load guesses again
        }
        pending_.assign(pc.points.begin(), pc.points.end());
        pending_kind_ = PENDING_LOAD;
        pending_range_ = range_limit;
        have_origin_ = false;
    }
    // NDTMap::loadPointCloudCentroid(cloud, origin, old_centroid, map_size, range_limit)
    // (fuser_hmt.cpp:201-202, 216-217; ndt_odom_debug.cpp:191): grid centre snapped to the old centroid's lattice so that
    // the cell faces of the two maps coincide; the range is measured from `origin`
    void loadPointCloudCentroid(const pcl::PointCloud<pcl::PointXYZ> &pc, const Eigen::Vector3d &origin,
                                const Eigen::Vector3d &old_centroid, const Eigen::Vector3d &map_size, double range_limit)
    {
        double c[3];
        for (int a = 0; a < 3; a++) {
            const double diff = origin(a) - old_centroid(a);
            c[a] = old_centroid(a) + std::floor(diff / res_) * res_;
            origin_[a] = origin(a);
        }
        set_geometry(c[0], c[1], c[2], map_size(0), map_size(1), map_size(2));
        pending_.assign(pc.points.begin(), pc.points.end());
        pending_kind_ = PENDING_LOAD;
        pending_range_ = range_limit;
        have_origin_ = true;
    }
    // NDTMap::addPointCloud(origin, cloud, classifierTh, maxz, sensor_noise, occupancy_limit)  (fuser_hmt.cpp:92, 485) on an
    // initialize()d map: the ray-traced insert.  classifierTh is unused upstream as well.
    void addPointCloud(const Eigen::Vector3d &origin, const pcl::PointCloud<pcl::PointXYZ> &pc, double /*classifierTh*/ = 0.06,
                       double maxz = 100.0, double sensor_noise = 0.25, double occupancy_limit = 255)
    {
        if (!initialized_) {     // isFirstLoad_: upstream falls back to loadPointCloud(pc)
            loadPointCloud(pc);
            return;
        }
        // Upstream a second addPointCloud before computeNDTCells adds its points and evidence to the same cells.  Here the
        // two calls are ONE device update (ray walk + moments + Gaussians) made at computeNDTCells: a second cloud from the
        // same origin with the same arguments joins the first; anything else would have to be dropped -- refuse it instead.
        if (pending_kind_ == PENDING_ADD && !pending_.empty()) {
            const bool same = origin_[0] == origin(0) && origin_[1] == origin(1) && origin_[2] == origin(2) && add_maxz_ == maxz &&
                              add_noise_ == sensor_noise && add_occ_limit_ == occupancy_limit;
            if (!same)
                throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTMap::addPointCloud: a cloud from another origin (or with other arguments) is "
                                                             "pending; call computeNDTCells between the two");
            pending_.insert(pending_.end(), pc.points.begin(), pc.points.end());
            return;
        }
        for (int a = 0; a < 3; a++) origin_[a] = origin(a);
        pending_.assign(pc.points.begin(), pc.points.end());
        pending_kind_ = PENDING_ADD;
        add_maxz_ = maxz;
        add_noise_ = sensor_noise;
        add_occ_limit_ = occupancy_limit;
    }
    // NDTMap::computeNDTCells(mode, maxnumpoints, occupancy_limit, origin, sensor_noise)  (fuser_hmt.cpp:94, 227, 486): the cells
    // of the pending cloud become Gaussians on the device.  origin / sensor_noise are unused upstream on the
    // SAMPLE_VARIANCE path.
    void computeNDTCells(int mode = CELL_UPDATE_MODE_SAMPLE_VARIANCE, unsigned maxnumpoints = 1e9, float occupancy_limit = 255,
                         const Eigen::Vector3d & /*origin*/ = Eigen::Vector3d(0, 0, 0), double /*sensor_noise*/ = 0.1)
    {
        if (mode != CELL_UPDATE_MODE_SAMPLE_VARIANCE)
            throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTMap::computeNDTCells: only CELL_UPDATE_MODE_SAMPLE_VARIANCE is implemented");
        if (pending_kind_ == PENDING_NONE) return;
        const void *pts = pending_.empty() ? nullptr : &pending_[0];
        if (pending_kind_ == PENDING_LOAD) {
            // a fresh map: every cell gets its first Gaussian, maxnumpoints cannot bind; the occupancy clamp is the default
            if (occupancy_limit != 255.f)
                throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTMap::computeNDTCells: occupancy_limit != 255 after loadPointCloud is not implemented");
            ndtgpu_cell_params cp;
            ndtgpu_default_cell_params(&cp);
            ndtgpu_host::check(ndtgpu_mapset_build_host(handle(), slot_, 1, pts, pending_.size(), sizeof(pcl::PointXYZ), 0,
                                                        pending_range_, have_origin_ ? origin_ : nullptr, &cp),
                               "ndtgpu_mapset_build_host");
        } else {
            ndtgpu_fuse_params fp;
            ndtgpu_default_fuse_params(&fp);
            fp.maxz = add_maxz_;
            fp.sensor_noise = add_noise_;
            if ((double)occupancy_limit != add_occ_limit_)
                throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTMap: addPointCloud and computeNDTCells must use the same occupancy_limit");
            fp.occupancy_limit = occupancy_limit;
            fp.maxnumpoints = (double)maxnumpoints;
            ndtgpu_host::check(ndtgpu_mapset_add_cloud_host(handle(), slot_, 1, pts, pending_.size(), sizeof(pcl::PointXYZ), 0, origin_, &fp),
                               "ndtgpu_mapset_add_cloud_host");
        }
        pending_.clear();
        pending_kind_ = PENDING_NONE;
    }
    int numberOfActiveCells()
    {
        uint32_t n = 0;
        ndtgpu_host::check(ndtgpu_mapset_num_cells(handle(), slot_, &n), "num_cells");
        return (int)n;
    }
    // NDTMap::getAllCells(): heap copies, the caller deletes (ndtgraph_conversion.h:34-43)
    std::vector<NDTCell *> getAllCells() { return cells_transformed(nullptr); }
    // NDTMap::pseudoTransformNDT(T) (fusion.h:840-841): transformed heap copies of the Gaussian cells, the caller deletes
    // (fusion.h:953-962, 1122-1131)
    std::vector<NDTCell *> pseudoTransformNDT(const Eigen::Affine3d &T) { return cells_transformed(&T); }
    // what ndt_feature::discardCell does through NDTMap::getCellAtPoint + hasGaussian_ = false (utils.h:229-236)
    bool discardCellAtPoint(const pcl::PointXYZ &pt)
    {
        const int before = numberOfActiveCells();
        const float xyz[3] = {pt.x, pt.y, pt.z};
        ndtgpu_host::check(ndtgpu_mapset_discard_cells(handle(), slot_, xyz, 1), "ndtgpu_mapset_discard_cells");
        return numberOfActiveCells() < before;
    }
    bool getCentroid(double &cx, double &cy, double &cz) const { cx = centre_[0]; cy = centre_[1]; cz = centre_[2]; return true; }
    // NDTCell::getOccupancy of every cell, slot order (x-major, y, z) -- needs initialize()
    std::vector<float> getOccupancy()
    {
        int32_t cpa[3];
        ndtgpu_host::check(ndtgpu_mapset_info(handle(), nullptr, cpa, nullptr), "mapset_info");
        std::vector<float> occ((size_t)cpa[0] * cpa[1] * cpa[2]);
        ndtgpu_host::check(ndtgpu_mapset_export_occupancy(handle(), slot_, occ.data()), "export_occupancy");
        return occ;
    }

    ndtgpu_mapset *handle()
    {
        ensure_set();
        return pool_ ? pool_->handle() : own_->handle();
    }
    size_t slot() const { return slot_; }
    double resolution() const { return res_; }

private:
    enum { PENDING_NONE, PENDING_LOAD, PENDING_ADD };
    void set_geometry(double cx, double cy, double cz, double sx, double sy, double sz)
    {
        const bool resize = sx != size_[0] || sy != size_[1] || sz != size_[2];
        if (pool_ && resize) throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTMap: a pooled map has the pool's extent");
        centre_[0] = cx; centre_[1] = cy; centre_[2] = cz;
        size_[0] = sx; size_[1] = sy; size_[2] = sz;
        have_geometry_ = have_size_ = true;
        if (own_ && resize) own_.reset();          // another extent: another device grid (the old content is gone, like upstream)
        ensure_set();
        // the device bins around the centre the host reports
        ndtgpu_host::check(ndtgpu_mapset_set_centre(pool_ ? pool_->handle() : own_->handle(), slot_, centre_), "set_centre");
    }
    void ensure_set()
    {
        if (pool_ || own_) return;
        if (!have_size_) throw ndtgpu_host::Error(NDTGPU_ERR_INVALID, "NDTMap: no geometry yet (initialize / guessSize / loadPointCloud first)");
        own_ = std::make_shared<ndtgpu_host::MapPool>(res_, centre_, size_, 1);
        slot_ = 0;
    }
    std::vector<NDTCell *> cells_transformed(const Eigen::Affine3d *T)
    {
        const uint32_t n = (uint32_t)numberOfActiveCells();
        std::vector<double> mean(3 * (size_t)n), cov(9 * (size_t)n);
        std::vector<int32_t> idx(3 * (size_t)n);
        std::vector<uint32_t> np(n);
        ndtgpu_host::check(ndtgpu_mapset_export_cells(handle(), slot_, mean.data(), cov.data(), idx.data(), np.data()), "export_cells");
        std::vector<NDTCell *> out(n);
        const double *m = T ? T->data() : nullptr;
        for (uint32_t i = 0; i < n; i++) {
            NDTCell *c = new NDTCell();
            Eigen::Vector3d mu(mean[3 * i], mean[3 * i + 1], mean[3 * i + 2]);
            Eigen::Matrix3d C;
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) C(r, k) = cov[9 * i + 3 * r + k];
            if (m) {
                Eigen::Vector3d t;
                Eigen::Matrix3d R;
                for (int r = 0; r < 3; r++) {
                    t(r) = m[0 * 4 + r] * mu(0) + m[1 * 4 + r] * mu(1) + m[2 * 4 + r] * mu(2) + m[12 + r];
                    for (int k = 0; k < 3; k++) R(r, k) = m[k * 4 + r];
                }
                mu = t;
                C = R * C * R.transpose();
            }
            c->setMean(mu);
            c->setCov(C);
            c->setN((int)np[i]);
            for (int a = 0; a < 3; a++) c->idx[a] = idx[3 * i + a];
            out[i] = c;
        }
        return out;
    }

    std::shared_ptr<ndtgpu_host::MapPool> pool_, own_;
    size_t slot_ = 0;
    double res_;
    CellVector *cv_ = nullptr;
    bool cv_owned_ = false;
    double centre_[3] = {0, 0, 0}, size_[3] = {0, 0, 0}, origin_[3] = {0, 0, 0};
    bool have_geometry_ = false, have_size_ = false, have_origin_ = false, initialized_ = false;
    std::vector<pcl::PointXYZ> pending_;
    int pending_kind_ = PENDING_NONE;
    double pending_range_ = -1., add_maxz_ = 100., add_noise_ = 0.25, add_occ_limit_ = 255.;
};

// NDTMatcherD2D: public knobs as set at ndt_feature_graph.cpp:261-262 and fusion.h:811-814
class NDTMatcherD2D {
public:
    int n_neighbours = 2;
    int ITR_MAX = 30;
    double DELTA_SCORE = 1e-3;   // default-constructed upstream: 10e-3 * current_resolution(0.1)  (include/ndtgpu.h, PROVENANCE)
    bool step_control = true;
    double lfd1 = 1.0, lfd2 = 0.05;
    int covariance_mode = 0;     // ndtgpu_covariance_batch `mode`
    ndtgpu_match_result last_result{};

    // bool match(NDTMap& target, NDTMap& source, Affine3d& T, bool useInitialGuess)  (graph.cpp:273)
    bool match(NDTMap &target, NDTMap &source, Eigen::Affine3d &T, bool useInitialGuess = false) { return match_dof(target, source, T, useInitialGuess, 0x3f); }

    // double derivativesNDT(sourceCells, targetNDT, score_gradient (6x1), Hessian (6x6), computeHessian)  (fusion.h:856, 1085;
    // the line searches :80, 238, 444, 617): ONE device evaluation per call
    double derivativesNDT(const std::vector<NDTCell *> &sourceNDT, NDTMap &targetNDT, Eigen::MatrixXd &score_gradient,
                          Eigen::MatrixXd &Hessian, bool computeHessian)
    {
        const size_t n = sourceNDT.size();
        mean_.resize(3 * n);
        cov_.resize(9 * n);
        for (size_t i = 0; i < n; i++) {
            const Eigen::Vector3d m = sourceNDT[i]->getMean();
            const Eigen::Matrix3d C = sourceNDT[i]->getCov();
            for (int a = 0; a < 3; a++) {
                mean_[3 * i + a] = m(a);
                for (int b = 0; b < 3; b++) cov_[9 * i + 3 * a + b] = C(a, b);
            }
        }
        double score = 0, g[6], H[36];
        ndtgpu_host::check(ndtgpu_derivatives(targetNDT.handle(), targetNDT.slot(), mean_.data(), cov_.data(), n, n_neighbours,
                                              computeHessian ? 1 : 0, lfd1, lfd2, &score, g, H), "ndtgpu_derivatives");
        if (score_gradient.rows() != 6 || score_gradient.cols() != 1) score_gradient.resize(6, 1);
        for (int a = 0; a < 6; a++) score_gradient(a, 0) = g[a];
        if (computeHessian) {
            if (Hessian.rows() != 6 || Hessian.cols() != 6) Hessian.resize(6, 6);
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) Hessian(a, b) = H[a * 6 + b];
        }
        return score;
    }

    // double lineSearchMT(increment, sourceCells, targetNDT)  (fusion.h:1013): More-Thuente step length along `increment`
    // (which may be negated in place, fusion.h:456-479); restated from the in-repo driver fusion.h:390-793 on top of
    // derivativesNDT.  The source cells are left untouched (trial copies are transformed, like upstream).
    template <class Vec6>
    double lineSearchMT(Vec6 &increment, std::vector<NDTCell *> &sourceNDT, NDTMap &targetNDT)
    {
        const double stpmax = 4.0, stpmin = 0.001, ftol = 0.11111, gtol = 0.99999, xtol = 0.01, recoverystep = 0.1;
        const int maxfev = 40;
        Eigen::MatrixXd g(6, 1), Hd(6, 6);
        const double finit = derivativesNDT(sourceNDT, targetNDT, g, Hd, false);
        double dginit = 0;
        for (int a = 0; a < 6; a++) dginit += increment(a) * g(a, 0);
        if (dginit >= 0.0) {
            for (int a = 0; a < 6; a++) increment(a) = -increment(a);
            dginit = -dginit;
            if (dginit >= 0.0) return recoverystep;
        }
        std::vector<NDTCell *> trial(sourceNDT.size());
        for (size_t i = 0; i < trial.size(); i++) trial[i] = new NDTCell();
        struct Free { std::vector<NDTCell *> &v; ~Free() { for (auto *c : v) delete c; } } guard{trial};
        auto phi = [&](double stp, double &dg) {
            const Eigen::Affine3d ps = ndtgpu_host::affine_from_pose(stp * increment(0), stp * increment(1), stp * increment(2),
                                                                     stp * increment(3), stp * increment(4), stp * increment(5));
            const Eigen::Matrix3d R = ps.rotation();
            for (size_t i = 0; i < trial.size(); i++) {
                trial[i]->setMean(ps * sourceNDT[i]->getMean());
                trial[i]->setCov(R * sourceNDT[i]->getCov() * R.transpose());
            }
            const double f = derivativesNDT(trial, targetNDT, g, Hd, false);
            dg = 0;
            for (int a = 0; a < 6; a++) dg += increment(a) * g(a, 0);
            return f;
        };
        double stp = 1.0, stx = 0.0, fx = finit, dgx = dginit, sty = 0.0, fy = finit, dgy = dginit, stmin, stmax;
        const double dgtest = ftol * dginit;
        double width = stpmax - stpmin, width1 = 2 * width;
        int infoc = 1, nfev = 0;
        bool brackt = false, stage1 = true;
        for (;;) {
            if (brackt) { stmin = std::fmin(stx, sty); stmax = std::fmax(stx, sty); }
            else { stmin = stx; stmax = stp + 4 * (stp - stx); }
            stp = std::fmax(stp, stpmin);
            stp = std::fmin(stp, stpmax);
            if ((brackt && ((stp <= stmin) || (stp >= stmax))) || (nfev >= maxfev - 1) || (infoc == 0) ||
                (brackt && (stmax - stmin <= xtol * stmax)))
                stp = stx;
            double dg = 0;
            const double f = phi(stp, dg);
            nfev++;
            const double ftest1 = finit + stp * dgtest;
            int info = 0;
            if ((brackt && ((stp <= stmin) || (stp >= stmax))) || (infoc == 0)) info = 6;
            if ((stp == stpmax) && (f <= ftest1) && (dg <= dgtest)) info = 5;
            if ((stp == stpmin) && ((f > ftest1) || (dg >= dgtest))) info = 4;
            if (nfev >= maxfev) info = 3;
            if (brackt && (stmax - stmin <= xtol * stmax)) info = 2;
            if ((f <= ftest1) && (std::fabs(dg) <= gtol * (-dginit))) info = 1;
            if (info != 0) return info == 1 ? stp : recoverystep;
            if (stage1 && (f <= ftest1) && (dg >= std::fmin(ftol, gtol) * dginit)) stage1 = false;
            if (stage1 && (f <= fx) && (f > ftest1)) {
                double fm = f - stp * dgtest, fxm = fx - stx * dgtest, fym = fy - sty * dgtest;
                double dgm = dg - dgtest, dgxm = dgx - dgtest, dgym = dgy - dgtest;
                infoc = cstep(stx, fxm, dgxm, sty, fym, dgym, stp, fm, dgm, brackt, stmin, stmax);
                fx = fxm + stx * dgtest; fy = fym + sty * dgtest; dgx = dgxm + dgtest; dgy = dgym + dgtest;
            } else {
                infoc = cstep(stx, fx, dgx, sty, fy, dgy, stp, f, dg, brackt, stmin, stmax);
            }
            if (brackt) {
                if (std::fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
                width1 = width;
                width = std::fabs(sty - stx);
            }
        }
    }

    // bool covariance(target, source, T, cov)  (graph.cpp:296-298; fuser_hmt.cpp:403-405)
    bool covariance(NDTMap &target, NDTMap &source, Eigen::Affine3d &T, Eigen::MatrixXd &cov)
    {
        uint32_t ti = (uint32_t)target.slot(), si = (uint32_t)source.slot();
        ndtgpu_match_params p = params(0x3f, true);
        double c36[36];
        int32_t singular = 0;
        ndtgpu_host::check(ndtgpu_covariance_batch(target.handle(), &ti, source.handle(), &si, T.data(), 1, &p, covariance_mode, c36,
                                                   &singular, nullptr), "ndtgpu_covariance_batch");
        if (cov.rows() != 6 || cov.cols() != 6) cov.resize(6, 6);
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) cov(a, b) = c36[a * 6 + b];
        return singular == 0;
    }

    ndtgpu_match_params params(int dof_mask, bool useInitialGuess) const
    {
        ndtgpu_match_params p;
        ndtgpu_default_match_params(&p);
        p.n_neighbours = n_neighbours;
        p.itr_max = ITR_MAX;
        p.delta_score = DELTA_SCORE;
        p.step_control = step_control ? 1 : 0;
        p.lfd1 = lfd1;
        p.lfd2 = lfd2;
        p.dof_mask = dof_mask;
        p.use_initial_guess = useInitialGuess ? 1 : 0;
        return p;
    }

protected:
    bool match_dof(NDTMap &target, NDTMap &source, Eigen::Affine3d &T, bool useInitialGuess, int dof_mask)
    {
        ndtgpu_match_params p = params(dof_mask, useInitialGuess);
        ndtgpu_host::check(ndtgpu_match_d2d(target.handle(), target.slot(), source.handle(), source.slot(), T.data(), &p, &last_result),
                           "ndtgpu_match_d2d");
        return last_result.converged != 0;
    }
    // MoreThuente::cstep == MINPACK-2 dcstep (More & Thuente, ACM TOMS 20(3), 1994); call sites fusion.h:756, 775
    static int cstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp, double fp, double dp,
                     bool &brackt, double stmin, double stmax)
    {
        int info = 0;
        bool bound;
        double theta, s, gamma, p, q, r, stpc, stpq, stpf;
        if ((brackt && ((stp <= std::fmin(stx, sty)) || (stp >= std::fmax(stx, sty)))) || (dx * (stp - stx) >= 0.0) || (stmax < stmin)) return info;
        const double sgnd = dp * (dx / std::fabs(dx));
        auto amax3 = [](double a, double b, double c) { return std::fmax(std::fmax(std::fabs(a), std::fabs(b)), std::fabs(c)); };
        if (fp > fx) {
            info = 1; bound = true;
            theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
            s = amax3(theta, dx, dp);
            gamma = s * std::sqrt(((theta / s) * (theta / s)) - (dx / s) * (dp / s));
            if (stp < stx) gamma = -gamma;
            p = (gamma - dx) + theta; q = ((gamma - dx) + gamma) + dp; r = p / q;
            stpc = stx + r * (stp - stx);
            stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2) * (stp - stx);
            stpf = (std::fabs(stpc - stx) < std::fabs(stpq - stx)) ? stpc : stpc + (stpq - stpc) / 2;
            brackt = true;
        } else if (sgnd < 0.0) {
            info = 2; bound = false;
            theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
            s = amax3(theta, dx, dp);
            gamma = s * std::sqrt(((theta / s) * (theta / s)) - (dx / s) * (dp / s));
            if (stp > stx) gamma = -gamma;
            p = (gamma - dp) + theta; q = ((gamma - dp) + gamma) + dx; r = p / q;
            stpc = stp + r * (stx - stp);
            stpq = stp + (dp / (dp - dx)) * (stx - stp);
            stpf = (std::fabs(stpc - stp) > std::fabs(stpq - stp)) ? stpc : stpq;
            brackt = true;
        } else if (std::fabs(dp) < std::fabs(dx)) {
            info = 3; bound = true;
            theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
            s = amax3(theta, dx, dp);
            gamma = s * std::sqrt(std::fmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
            if (stp > stx) gamma = -gamma;
            p = (gamma - dp) + theta; q = (gamma + (dx - dp)) + gamma; r = p / q;
            if ((r < 0.0) && (gamma != 0.0)) stpc = stp + r * (stx - stp);
            else if (stp > stx) stpc = stmax;
            else stpc = stmin;
            stpq = stp + (dp / (dp - dx)) * (stx - stp);
            if (brackt) stpf = (std::fabs(stp - stpc) < std::fabs(stp - stpq)) ? stpc : stpq;
            else stpf = (std::fabs(stp - stpc) > std::fabs(stp - stpq)) ? stpc : stpq;
        } else {
            info = 4; bound = false;
            if (brackt) {
                theta = 3 * (fp - fy) / (sty - stp) + dy + dp;
                s = amax3(theta, dy, dp);
                gamma = s * std::sqrt(((theta / s) * (theta / s)) - (dy / s) * (dp / s));
                if (stp > sty) gamma = -gamma;
                p = (gamma - dp) + theta; q = ((gamma - dp) + gamma) + dy; r = p / q;
                stpc = stp + r * (sty - stp);
                stpf = stpc;
            } else if (stp > stx) stpf = stmax;
            else stpf = stmin;
        }
        if (fp > fx) { sty = stp; fy = fp; dy = dp; }
        else {
            if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
            stx = stp; fx = fp; dx = dp;
        }
        stpf = std::fmin(stmax, stpf);
        stpf = std::fmax(stmin, stpf);
        stp = stpf;
        if (brackt && bound) {
            if (sty > stx) stp = std::fmin(stx + 0.66 * (sty - stx), stp);
            else stp = std::fmax(stx + 0.66 * (sty - stx), stp);
        }
        return info;
    }
    std::vector<double> mean_, cov_;   // staging for derivativesNDT
};

// NDTMatcherD2D_2D (fusion.h:1170-1175): {x, y, yaw}
class NDTMatcherD2D_2D : public NDTMatcherD2D {
public:
    bool match(NDTMap &target, NDTMap &source, Eigen::Affine3d &T, bool useInitialGuess = false) { return match_dof(target, source, T, useInitialGuess, 0x23); }
};

}  // namespace lslgeneric

namespace ndtgpu_host {

// (added; no counterpart class upstream) Scans in, poses out for MANY pairs in ONE call: what the reference does pair after pair
// -- a NDTMap per scan through loadPointCloud + computeNDTCells (fuser_hmt.cpp:195-227), then NDTMatcherD2D::match
// (graph.cpp:273; the loop graph.cpp:347-353) -- as one ndtgpu_register_batch_host: the grid builds of all scans of a sub-batch are
// one launch, their registrations one launch, sub-batches pipelined over the library's streams.  Pair k registers moving[k]
// (source) against fixed[k] (target); T[k] holds the initial guess and receives the pose.  Clouds shorter than the longest
// are padded with NaN points, which loadPointCloud drops.
class ScanRegistrar {
public:
    ScanRegistrar(double res, const double centre[3], const double size_m[3], size_t pairs_per_batch = 1024, int depth = 8,
                  uint32_t max_cells = 0)
    {
        ndtgpu_grid_params g;
        g.res = res;
        for (int a = 0; a < 3; a++) { g.centre[a] = centre[a]; g.size[a] = size_m[a]; }
        g.max_cells = max_cells;
        check(ndtgpu_registrar_create(&g, pairs_per_batch, depth, &reg_), "ndtgpu_registrar_create");
    }
    ~ScanRegistrar() { ndtgpu_registrar_destroy(reg_); }
    ScanRegistrar(const ScanRegistrar &) = delete;
    ScanRegistrar &operator=(const ScanRegistrar &) = delete;

    // returns match()'s return value per pair; `results` (optional) receives what the device matcher reported
    std::vector<bool> match(const lslgeneric::NDTMatcherD2D &matcher, const std::vector<pcl::PointCloud<pcl::PointXYZ>> &fixed,
                            const std::vector<pcl::PointCloud<pcl::PointXYZ>> &moving, std::vector<Eigen::Affine3d> &T,
                            double range_limit = -1., bool useInitialGuess = true, std::vector<ndtgpu_match_result> *results = nullptr,
                            int dof_mask = 0x3f)
    {
        const size_t n = fixed.size();
        if (moving.size() != n || T.size() != n) throw Error(NDTGPU_ERR_INVALID, "ScanRegistrar::match: one moving cloud and one pose per fixed cloud");
        size_t np = 0;
        for (size_t k = 0; k < n; k++) np = std::max(np, std::max(fixed[k].size(), moving[k].size()));
        static_assert(sizeof(pcl::PointXYZ) == 16, "pcl::PointXYZ is four floats");
        const float nan = std::nanf("");
        pts_.assign(2 * n * np * 4, nan);
        for (size_t k = 0; k < n; k++) {
            for (size_t i = 0; i < fixed[k].size(); i++) {
                const pcl::PointXYZ &q = fixed[k].points[i];
                float *o = &pts_[(k * np + i) * 4];
                o[0] = q.x; o[1] = q.y; o[2] = q.z;
            }
            for (size_t i = 0; i < moving[k].size(); i++) {
                const pcl::PointXYZ &q = moving[k].points[i];
                float *o = &pts_[((n + k) * np + i) * 4];
                o[0] = q.x; o[1] = q.y; o[2] = q.z;
            }
        }
        std::vector<double> T16(16 * n);
        for (size_t k = 0; k < n; k++)
            for (int e = 0; e < 16; e++) T16[16 * k + e] = T[k].data()[e];
        std::vector<ndtgpu_match_result> res(n);
        ndtgpu_match_params p = matcher.params(dof_mask, useInitialGuess);
        check(ndtgpu_register_batch_host(reg_, pts_.data(), pts_.data() + n * np * 4, np, 16, np * 16, range_limit, nullptr, T16.data(), n, &p,
                                         res.data()), "ndtgpu_register_batch_host");
        std::vector<bool> ok(n);
        for (size_t k = 0; k < n; k++) {
            for (int e = 0; e < 16; e++) T[k].data()[e] = T16[16 * k + e];
            ok[k] = res[k].converged != 0;
        }
        if (results) *results = res;
        return ok;
    }

private:
    ndtgpu_registrar *reg_ = nullptr;
    std::vector<float> pts_;
};

}  // namespace ndtgpu_host
