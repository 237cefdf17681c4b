// lslgeneric_gpu.h -- host C++ mirror of the lslgeneric:: classes that MalcolmMielle/ndt_feature_graph
// calls on its hot path, implemented over the C-ABI of libndtgpu.so (include/ndtgpu.h).
//
// Same class and member names, argument meaning and error behaviour as the call sites in
//   ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:87-94, 195-227
//   ndt_feature/src/ndt_feature_src/ndt_feature_graph.cpp:261-273
//   ndt_feature/include/ndt_feature/ndt_matcher_d2d_fusion.h:811-814, 840, 856, 1170-1175
//   ndt_feature/src/ndt_odom_debug.cpp:163-206
// so that a maintainer can point those translation units at this header instead of
// <ndt_map/ndt_map.h>, <ndt_map/lazy_grid.h>, <ndt_registration/ndt_matcher_d2d.h>.
// Everything that computes lives on the GPU; there is no CPU fallback: a failed C-ABI call throws
// ndtgpu_host::Error (the reference has no error channel here besides bool returns).
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
        : n_maps_(n_maps)
    {
        ndtgpu_grid_params g;
        g.res = res;
        for (int a = 0; a < 3; a++) { g.centre[a] = centre[a]; g.size[a] = size_m[a]; }
        g.max_cells = max_cells;
        check(ndtgpu_mapset_create(&g, n_maps, &set_), "ndtgpu_mapset_create");
    }
    ~MapPool() { ndtgpu_mapset_destroy(set_); }
    MapPool(const MapPool &) = delete;
    MapPool &operator=(const MapPool &) = delete;
    ndtgpu_mapset *handle() const { return set_; }
    size_t size() const { return n_maps_; }
    size_t allocate()
    {
        if (next_ >= n_maps_) throw Error(NDTGPU_ERR_CAPACITY, "MapPool exhausted");
        return next_++;
    }

private:
    ndtgpu_mapset *set_ = nullptr;
    size_t n_maps_, next_ = 0;
};

}  // namespace ndtgpu_host

namespace lslgeneric {

using ndtgpu_host::Affine3d;
using ndtgpu_host::PointCloud;
using ndtgpu_host::PointXYZ;

enum NDTCellUpdateMode { CELL_UPDATE_MODE_COVARIANCE_INTERSECTION, CELL_UPDATE_MODE_SAMPLE_VARIANCE };

// read-only view of one Gaussian (NDTCell::getMean / getCov / hasGaussian_)
class NDTCell {
public:
    bool hasGaussian_ = true;
    std::array<double, 3> mean{};
    std::array<double, 9> cov{};    // row-major
    std::array<int, 3> idx{};
    unsigned npts = 0;
    const std::array<double, 3> &getMean() const { return mean; }
    const std::array<double, 9> &getCov() const { return cov; }
    void setMean(const std::array<double, 3> &m) { mean = m; }
    void setCov(const std::array<double, 9> &c) { cov = c; }
};

// LazyGrid(res): only carries the cell size (the dense table lives on the device)
class LazyGrid {
public:
    explicit LazyGrid(double cellSize) : res(cellSize) {}
    double res;
};
using SpatialIndex = LazyGrid;

class NDTMap {
public:
    // new NDTMap(new LazyGrid(res))  -- takes ownership of idx like the reference (fuser_hmt.cpp:87)
    explicit NDTMap(SpatialIndex *idx, bool /*dealloc*/ = false) : res_(idx->res) { delete idx; }
    // a map that lives in a shared pool (graph node maps): geometry comes from the pool
    NDTMap(std::shared_ptr<ndtgpu_host::MapPool> pool, size_t slot) : pool_(std::move(pool)), slot_(slot), res_(0) {}

    // NDTMap::initialize(cx,cy,cz,sx,sy,sz)  (fuser_hmt.cpp:89)
    void initialize(double cx, double cy, double cz, double sx, double sy, double sz) { guessSize(cx, cy, cz, sx, sy, sz); }
    // NDTMap::guessSize(cx,cy,cz,sx,sy,sz)  (fuser_hmt.cpp:222): explicit centre + extent
    void guessSize(double cx, double cy, double cz, double sx, double sy, double sz)
    {
        centre_[0] = cx; centre_[1] = cy; centre_[2] = cz;
        size_[0] = sx; size_[1] = sy; size_[2] = sz;
        have_geometry_ = true;
        if (pool_) ndtgpu_host::check(ndtgpu_mapset_set_centre(pool_->handle(), slot_, centre_), "set_centre");
    }
    // NDTMap::setMapSize (ndt_odom_debug.cpp:177): extent only, centre = centroid of the cloud
    void setMapSize(double sx, double sy, double sz) { size_[0] = sx; size_[1] = sy; size_[2] = sz; have_size_ = true; }

    // NDTMap::loadPointCloud(cloud, range_limit)  (fuser_hmt.cpp:225; ndt_odom_debug.cpp:178)
    void loadPointCloud(const PointCloud<PointXYZ> &pc, double range_limit = -1.)
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
            for (int a = 0; a < 3; a++) centre_[a] = c[a];
            if (!have_size_) { size_[0] = size_[1] = 4 * maxd; size_[2] = 4 * maxd; }
        }
        ensure_set();
        pending_ = &pc;
        pending_range_ = range_limit;
        have_origin_ = false;
    }
    // NDTMap::loadPointCloudCentroid(cloud, origin, old_centroid, map_size, range_limit)
    // (fuser_hmt.cpp:201-217; ndt_odom_debug.cpp:191): grid centre snapped to the old centroid's lattice
    void loadPointCloudCentroid(const PointCloud<PointXYZ> &pc, const double origin[3], const double old_centroid[3],
                                const double map_size[3], double range_limit)
    {
        for (int a = 0; a < 3; a++) {
            double diff = origin[a] - old_centroid[a];
            centre_[a] = old_centroid[a] + std::floor(diff / res()) * res();
            size_[a] = map_size[a];
            origin_[a] = origin[a];
        }
        have_geometry_ = true;
        ensure_set();
        pending_ = &pc;
        pending_range_ = range_limit;
        have_origin_ = true;
    }
    // NDTMap::computeNDTCells(mode, maxnumpoints, occupancy_limit, origin, sensor_noise)
    // (fuser_hmt.cpp:94, 227): grid build + Gaussians on the device
    void computeNDTCells(int /*mode*/ = CELL_UPDATE_MODE_SAMPLE_VARIANCE, unsigned /*maxnumpoints*/ = 100000,
                         float /*occupancy_limit*/ = 255, const double * /*origin*/ = nullptr, double /*noise*/ = 0.1)
    {
        if (!pending_) return;
        ndtgpu_cell_params cp;
        ndtgpu_default_cell_params(&cp);
        const auto &pts = pending_->points;
        ndtgpu_host::check(ndtgpu_mapset_build_host(handle(), slot_, 1, pts.empty() ? nullptr : &pts[0], pts.size(),
                                                    sizeof(PointXYZ), 0, pending_range_, have_origin_ ? origin_ : nullptr, &cp),
                           "ndtgpu_mapset_build_host");
        pending_ = nullptr;
    }
    int numberOfActiveCells()
    {
        uint32_t n = 0;
        ndtgpu_host::check(ndtgpu_mapset_num_cells(handle(), slot_, &n), "num_cells");
        return (int)n;
    }
    std::vector<NDTCell> getAllCells()
    {
        uint32_t n = (uint32_t)numberOfActiveCells();
        std::vector<double> mean(3 * n), cov(9 * n);
        std::vector<int32_t> idx(3 * n);
        std::vector<uint32_t> np(n);
        ndtgpu_host::check(ndtgpu_mapset_export_cells(handle(), slot_, mean.data(), cov.data(), idx.data(), np.data()), "export_cells");
        std::vector<NDTCell> out(n);
        for (uint32_t i = 0; i < n; i++) {
            for (int a = 0; a < 3; a++) { out[i].mean[a] = mean[3 * i + a]; out[i].idx[a] = idx[3 * i + a]; }
            for (int a = 0; a < 9; a++) out[i].cov[a] = cov[9 * i + a];
            out[i].npts = np[i];
        }
        return out;
    }
    // NDTMap::pseudoTransformNDT(T) (fusion.h:840): transformed copies of the Gaussian cells
    std::vector<NDTCell> pseudoTransformNDT(const Affine3d &T)
    {
        std::vector<NDTCell> cells = getAllCells();
        const double *m = ndtgpu_host::affine_data(T);
        for (auto &c : cells) {
            std::array<double, 3> mu{};
            for (int r = 0; r < 3; r++) mu[r] = m[0 * 4 + r] * c.mean[0] + m[1 * 4 + r] * c.mean[1] + m[2 * 4 + r] * c.mean[2] + m[12 + r];
            std::array<double, 9> rc{}, out{};
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++)
                    for (int j = 0; j < 3; j++) rc[r * 3 + k] += m[j * 4 + r] * c.cov[j * 3 + k];
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++)
                    for (int j = 0; j < 3; j++) out[r * 3 + k] += rc[r * 3 + j] * m[j * 4 + k];
            c.mean = mu;
            c.cov = out;
        }
        return cells;
    }
    void getCentroid(double &cx, double &cy, double &cz) const { cx = centre_[0]; cy = centre_[1]; cz = centre_[2]; }

    ndtgpu_mapset *handle()
    {
        ensure_set();
        return pool_ ? pool_->handle() : own_->handle();
    }
    size_t slot() const { return slot_; }

private:
    double res() const { return res_; }
    void ensure_set()
    {
        if (pool_ || own_) return;
        own_ = std::make_shared<ndtgpu_host::MapPool>(res_, centre_, size_, 1);
        slot_ = 0;
    }
    std::shared_ptr<ndtgpu_host::MapPool> pool_, own_;
    size_t slot_ = 0;
    double res_;
    double centre_[3] = {0, 0, 0}, size_[3] = {0, 0, 0}, origin_[3] = {0, 0, 0};
    bool have_geometry_ = false, have_size_ = false, have_origin_ = false;
    const PointCloud<PointXYZ> *pending_ = nullptr;
    double pending_range_ = -1.;
};

// NDTMatcherD2D: public knobs as set at ndt_feature_graph.cpp:261-262 and fusion.h:811-814
class NDTMatcherD2D {
public:
    int n_neighbours = 2;
    int ITR_MAX = 30;
    double DELTA_SCORE = 1e-3;   // default-constructed upstream: 10e-3 * current_resolution(0.1)  
    bool step_control = true;
    double lfd1 = 1.0, lfd2 = 0.05;
    ndtgpu_match_result last_result{};

    // bool match(NDTMap& target, NDTMap& source, Affine3d& T, bool useInitialGuess)  (graph.cpp:273)
    bool match(NDTMap &target, NDTMap &source, Affine3d &T, bool useInitialGuess = false)
    {
        ndtgpu_match_params p = params(0x3f, useInitialGuess);
        ndtgpu_host::check(ndtgpu_match_d2d(target.handle(), target.slot(), source.handle(), source.slot(),
                                            ndtgpu_host::affine_data(T), &p, &last_result), "ndtgpu_match_d2d");
        return last_result.converged != 0;
    }
    // double derivativesNDT(cells, targetNDT, score_gradient(6), Hessian(6x6 row-major), computeHessian)  (fusion.h:856)
    double derivativesNDT(const std::vector<NDTCell> &sourceNDT, NDTMap &targetNDT, double score_gradient[6],
                          double Hessian[36], bool computeHessian)
    {
        std::vector<double> mean(3 * sourceNDT.size()), cov(9 * sourceNDT.size());
        for (size_t i = 0; i < sourceNDT.size(); i++) {
            for (int a = 0; a < 3; a++) mean[3 * i + a] = sourceNDT[i].mean[a];
            for (int a = 0; a < 9; a++) cov[9 * i + a] = sourceNDT[i].cov[a];
        }
        double score = 0;
        ndtgpu_host::check(ndtgpu_derivatives(targetNDT.handle(), targetNDT.slot(), mean.data(), cov.data(), sourceNDT.size(),
                                              n_neighbours, computeHessian ? 1 : 0, lfd1, lfd2, &score, score_gradient, Hessian),
                           "ndtgpu_derivatives");
        return score;
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
};

// NDTMatcherD2D_2D (fusion.h:1170-1175): {x, y, yaw}
class NDTMatcherD2D_2D : public NDTMatcherD2D {
public:
    bool match(NDTMap &target, NDTMap &source, Affine3d &T, bool useInitialGuess = false)
    {
        ndtgpu_match_params p = params(0x23, useInitialGuess);
        ndtgpu_host::check(ndtgpu_match_d2d(target.handle(), target.slot(), source.handle(), source.slot(),
                                            ndtgpu_host::affine_data(T), &p, &last_result), "ndtgpu_match_d2d");
        return last_result.converged != 0;
    }
};

}  // namespace lslgeneric
