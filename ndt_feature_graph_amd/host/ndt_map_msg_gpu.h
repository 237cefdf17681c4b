// ndt_map_msg_gpu.h -- the cell-map wire format of the untouched ROS layer:
// lslgeneric::toMessage / fromMessage between an NDTMap on the GPU and ndt_map::NDTMapMsg, as called at
//   ndt_feature/include/ndt_feature/ndtgraph_conversion.h:34-43 (fuserHMTToMsg: toMessage(fuser.map, m.map, "/world"))
//   ndt_feature/include/ndt_feature/ndtgraph_conversion.h:129-158 (msgTofuserHMT: fromMessage(lz, fuser.map, m.map, frame, true))
// PROVENANCE: NDTMapMsg / NDTCellMsg are message definitions of perception_oru's ndt_map package (un-vendored; the
// reference only uses the type, msg/NDTFeatureFuserHMTMsg.msg:4).  The field lists below are restated from memory; in a
// catkin build include the generated <ndt_map/NDTMapMsg.h> instead of these plain structs (define NDTGPU_USE_ROS_MSGS)
// -- the conversion code only touches the fields named here.  The JFF disk format (fuser_hmt.cpp:20-49) is a raw dump of
// perception_oru's C++ objects and is not reproduced.
#pragma once
#include "lslgeneric_gpu.h"

#ifdef NDTGPU_USE_ROS_MSGS
#include <ndt_map/NDTMapMsg.h>
#else
namespace ndt_map {
struct NDTCellMsg {
    double mean_x = 0, mean_y = 0, mean_z = 0;
    double occupancy = 0;                 // NDTCell::getOccupancyRescaled()
    std::vector<double> cov_matrix;       // 9 values, row by row
    double N = 0;
    bool hasGaussian_ = false;
};
struct NDTMapMsg {
    struct { std::string frame_id; } header;
    double x_size = 0, y_size = 0, z_size = 0;                   // extent [m]
    double x_cen = 0, y_cen = 0, z_cen = 0;                      // centre [m]
    double x_cell_size = 0, y_cell_size = 0, z_cell_size = 0;    // cell size [m]
    std::vector<NDTCellMsg> cells;
};
}  // namespace ndt_map
#endif

namespace lslgeneric {

// NDTCell::getOccupancyRescaled in float arithmetic (expf restated as the correctly rounded float(exp(double)))
inline float occupancyRescaled(float occ)
{
    const float e = (float)std::exp((double)occ);
    const float o = 1.0f - 1.0f / (1.0f + e);
    return o > 1 ? 1 : (o < 0 ? 0 : o);
}

// bool toMessage(NDTMap *map, NDTMapMsg &msg, std::string frame_name): every cell that carries a reading (Gaussian
// cells with their mean / covariance / N; cells without a Gaussian with their occupancy only).  A map that never went
// through initialize() has readings exactly in its Gaussian cells.
inline bool toMessage(NDTMap *map, ndt_map::NDTMapMsg &msg, std::string frame_name)
{
    if (!map) return false;
    msg.header.frame_id = frame_name;
    int32_t cpa[3];
    size_t n_maps = 0;
    ndtgpu_host::check(ndtgpu_mapset_info(map->handle(), &n_maps, cpa, nullptr), "mapset_info");
    const double res = map->resolution();
    msg.x_cell_size = msg.y_cell_size = msg.z_cell_size = res;
    msg.x_size = cpa[0] * res; msg.y_size = cpa[1] * res; msg.z_size = cpa[2] * res;
    map->getCentroid(msg.x_cen, msg.y_cen, msg.z_cen);
    std::vector<NDTCell *> cells = map->getAllCells();
    std::vector<float> occ;
    bool have_occ = true;
    try { occ = map->getOccupancy(); } catch (const ndtgpu_host::Error &) { have_occ = false; }
    std::vector<char> sent(have_occ ? occ.size() : 0, 0);
    msg.cells.clear();
    for (NDTCell *c : cells) {
        ndt_map::NDTCellMsg m;
        const Eigen::Vector3d mu = c->getMean();
        const Eigen::Matrix3d C = c->getCov();
        m.mean_x = mu(0); m.mean_y = mu(1); m.mean_z = mu(2);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) m.cov_matrix.push_back(C(i, j));
        m.N = c->getN();
        m.hasGaussian_ = true;
        const size_t slot = ((size_t)c->idx[0] * cpa[1] + c->idx[1]) * cpa[2] + c->idx[2];
        m.occupancy = have_occ ? occupancyRescaled(occ[slot]) : occupancyRescaled((float)(m.N * 0.4054651081081642));
        if (have_occ) sent[slot] = 1;
        msg.cells.push_back(m);
        delete c;
    }
    if (have_occ)
        for (size_t s = 0; s < occ.size(); s++) {
            if (sent[s] || occ[s] == 0.0f) continue;
            ndt_map::NDTCellMsg m;                         // a cell with a reading but no Gaussian: its centre stands in for the mean
            const size_t iz = s % cpa[2], iy = (s / cpa[2]) % cpa[1], ix = s / ((size_t)cpa[2] * cpa[1]);
            m.mean_x = msg.x_cen + ((double)ix - cpa[0] / 2) * res;
            m.mean_y = msg.y_cen + ((double)iy - cpa[1] / 2) * res;
            m.mean_z = msg.z_cen + ((double)iz - cpa[2] / 2) * res;
            m.cov_matrix.assign(9, 0.0);
            m.occupancy = occupancyRescaled(occ[s]);
            msg.cells.push_back(m);
        }
    return true;
}

// bool fromMessage(LazyGrid *&idx, NDTMap *&map, NDTMapMsg msg, std::string &frame_name, bool dealloc): a new map with
// the message's geometry; Gaussian cells are installed as they are, occupancies are stored as log-odds of the rescaled
// value the message carries.
inline bool fromMessage(LazyGrid *&idx, NDTMap *&map, const ndt_map::NDTMapMsg &msg, std::string &frame_name, bool dealloc = false)
{
    if (!(msg.x_cell_size == msg.y_cell_size && msg.y_cell_size == msg.z_cell_size)) return false;   // cubic cells only
    idx = nullptr;                                       // the mirror's NDTMap owns its grid
    map = new NDTMap(new LazyGrid(msg.x_cell_size), dealloc);
    map->initialize(msg.x_cen, msg.y_cen, msg.z_cen, msg.x_size, msg.y_size, msg.z_size);
    frame_name = msg.header.frame_id;
    int32_t cpa[3];
    ndtgpu_host::check(ndtgpu_mapset_info(map->handle(), nullptr, cpa, nullptr), "mapset_info");
    std::vector<double> mean, cov;
    std::vector<float> occ((size_t)cpa[0] * cpa[1] * cpa[2], 0.0f);
    const double res = msg.x_cell_size, c[3] = {msg.x_cen, msg.y_cen, msg.z_cen};
    for (const auto &m : msg.cells) {
        const double p[3] = {m.mean_x, m.mean_y, m.mean_z};
        int id[3];
        bool inside = true;
        for (int a = 0; a < 3; a++) {
            id[a] = (int)(std::floor((p[a] - c[a]) / res + 0.5) + cpa[a] / 2.0);
            inside = inside && id[a] >= 0 && id[a] < cpa[a];
        }
        if (!inside) continue;
        const double q = std::fmin(std::fmax(m.occupancy, 1e-7), 1.0 - 1e-7);
        occ[((size_t)id[0] * cpa[1] + id[1]) * cpa[2] + id[2]] = (float)std::log(q / (1.0 - q));
        if (m.hasGaussian_ && m.cov_matrix.size() == 9) {
            mean.insert(mean.end(), p, p + 3);
            cov.insert(cov.end(), m.cov_matrix.begin(), m.cov_matrix.end());
        }
    }
    ndtgpu_host::check(ndtgpu_mapset_set_cells(map->handle(), map->slot(), mean.data(), cov.data(), mean.size() / 3), "ndtgpu_mapset_set_cells");
    ndtgpu_host::check(ndtgpu_mapset_import_occupancy(map->handle(), map->slot(), occ.data()), "ndtgpu_mapset_import_occupancy");
    return true;
}

}  // namespace lslgeneric
