// host_demo.cpp -- the reference's offline edge-refinement flow (ndt_feature_graph_opt.cpp:131-160:
// all possible links -> NDT registration -> gates) written against the host mirror, with the
// synthetic corridor of ndt_odom_debug.cpp:94-119 as input.  Exit code 0 = every check passed.
//   no GPU : checks that the library loads and fails loudly (NDTGPU_ERR_NO_DEVICE), exit 0
//   GPU    : builds N node maps, registers all pairs in one batch, checks the recovered poses
#include "ndt_feature_graph_gpu.h"

#include <cstdio>
#include <random>

using namespace ndt_feature;

static PointCloud<PointXYZ> corridor_scan(const Affine3d &sensor_pose_world, unsigned seed, int n_per_wall = 4000)
{
    // two walls y = +-2 and an end wall x = 12, seen from `sensor_pose_world`; points in the sensor frame
    std::mt19937 rng(seed);
    std::normal_distribution<double> nd(0.0, 0.03);
    std::uniform_real_distribution<double> uz(0.0, 0.02);
    PointCloud<PointXYZ> pc;
    Affine3d inv = sensor_pose_world.inverse();
    auto add = [&](double wx, double wy) {
        double x = inv(0, 0) * wx + inv(0, 1) * wy + inv(0, 3), y = inv(1, 0) * wx + inv(1, 1) * wy + inv(1, 3);
        pc.push_back(PointXYZ((float)(x + nd(rng)), (float)(y + nd(rng)), (float)uz(rng)));
    };
    for (int j = 0; j < n_per_wall; j++) {
        double t = -8.0 + 20.0 * j / n_per_wall;
        add(t, 2.0 + 0.3 * std::sin(0.9 * t));
        add(t, -2.0 - 0.2 * std::cos(0.7 * t));
    }
    for (int j = 0; j < n_per_wall / 4; j++) add(12.0, -2.0 + 4.0 * j / (n_per_wall / 4.0));
    return pc;
}

int main()
{
    if (ndtgpu_device_count() < 1) {
        NDTFeatureGraph::Params p;
        try {
            NDTFeatureGraph g(p);
            std::printf("FAIL: graph construction succeeded without a device\n");
            return 1;
        } catch (const ndtgpu_host::Error &e) {
            if (e.status != NDTGPU_ERR_NO_DEVICE) { std::printf("FAIL: wrong status %d\n", e.status); return 1; }
            std::printf("no GPU: %s -- OK (no CPU fallback)\n", e.what());
            return 0;
        }
    }
    NDTFeatureGraph::Params p;
    p.resolution = 0.5; p.map_size_x = 60; p.map_size_y = 60; p.map_size_z = 1; p.sensor_range = 30; p.max_nodes = 8;
    NDTFeatureGraph graph(p);
    const int N = 5;
    std::vector<Affine3d> gt;
    for (int k = 0; k < N; k++) {
        Affine3d T = Affine3d::fromPose(0.35 * k, 0.05 * k, 0, 0, 0, 0.02 * k);
        gt.push_back(T);
        // odometry-like node pose: ground truth perturbed
        Affine3d odo = Affine3d::fromPose(0.35 * k + 0.03 * (k % 2 ? 1 : -1), 0.05 * k - 0.02, 0, 0, 0, 0.02 * k + 0.004);
        graph.addNode(odo, corridor_scan(T, 100 + k));
    }
    int fails = 0;
    for (int k = 0; k < N; k++)
        if (graph.getMap(k)->numberOfActiveCells() < 20) { std::printf("FAIL: node %d has too few cells\n", k); fails++; }

    std::vector<NDTFeatureLink> links = graph.computeAllPossibleLinks();
    std::vector<NDTFeatureLink> serial = links;
    graph.updateLinksUsingNDTRegistration(links, 2, true);                 // one batched call
    for (auto &l : serial) graph.updateLinkUsingNDTRegistration(l, 2, true);   // the reference's loop shape
    for (size_t k = 0; k < links.size(); k++) {
        Affine3d want = gt[links[k].ref_idx].inverse() * gt[links[k].mov_idx];
        double d, a;
        distanceBetweenAffine3d(want, links[k].T, d, a);
        // the batched call runs the persistent kernel, a single link the host-driven multi-workgroup path:
        // same algorithm, different summation order
        bool same = true;
        for (int q = 0; q < 16; q++) same = same && std::fabs(links[k].T.m[q] - serial[k].T.m[q]) < 1e-9;
        std::printf("link %zu-%zu: |dt| %.4f m  |dyaw| %.5f rad  iters %d  converged %d  batch==single %d\n", links[k].ref_idx,
                    links[k].mov_idx, d, a, links[k].iterations, (int)links[k].converged, (int)same);
        if (d > 0.06 || a > 0.01 || !same) fails++;   // grid-limited accuracy with the edge preset (DELTA_SCORE 1e-3)
    }
    std::vector<NDTFeatureLink> valid = graph.getValidLinks(links, 1e9, 1.0, 0.2, 2);
    std::printf("%zu links, %zu valid after the gates, %d failures\n", links.size(), valid.size(), fails);
    // single-map API used by the fuser call sites
    lslgeneric::NDTMap ndglobal(new lslgeneric::LazyGrid(0.5), true);
    ndglobal.guessSize(0, 0, 0, 60, 60, 1);
    PointCloud<PointXYZ> pc = corridor_scan(gt[0], 100);
    ndglobal.loadPointCloud(pc, 30.0);
    ndglobal.computeNDTCells(lslgeneric::CELL_UPDATE_MODE_SAMPLE_VARIANCE);
    if (ndglobal.numberOfActiveCells() != graph.getMap(0)->numberOfActiveCells()) { std::printf("FAIL: stand-alone map differs\n"); fails++; }
    lslgeneric::NDTMatcherD2D m;
    m.n_neighbours = 2;
    Affine3d T = Affine3d::Identity();
    bool conv = m.match(*graph.getMap(0), ndglobal, T, true);
    double d, a;
    distanceBetweenAffine3d(Affine3d::Identity(), T, d, a);
    if (!conv || d > 1e-6) { std::printf("FAIL: self match moved by %g\n", d); fails++; }
    // matchFusion with a tight odometry prior stays at the initial guess, with a loose one it equals match()
    {
        Affine3d Tg = Affine3d::fromPose(0.05, -0.02, 0, 0, 0, 0.003), Ta = Tg, Tb = Tg, Tc = Tg;
        double tight[36] = {0}, loose[36] = {0};
        for (int q = 0; q < 6; q++) { tight[q * 7] = 1e-12; loose[q * 7] = 1e12; }
        matchFusion(*graph.getMap(0), ndglobal, Ta, tight, true, true, 30, 2, 1e-6, true);
        matchFusion(*graph.getMap(0), ndglobal, Tb, loose, true, true, 30, 2, 1e-6, true);
        lslgeneric::NDTMatcherD2D m2; m2.n_neighbours = 2; m2.ITR_MAX = 30; m2.DELTA_SCORE = 1e-6;
        m2.match(*graph.getMap(0), ndglobal, Tc, true);
        double d1, a1, d2, a2;
        distanceBetweenAffine3d(Tg, Ta, d1, a1);
        distanceBetweenAffine3d(Tc, Tb, d2, a2);
        std::printf("matchFusion: tight prior moved %.2e m, loose prior differs from match() by %.2e m\n", d1, d2);
        if (d1 > 1e-6 || d2 > 1e-6) fails++;
    }
    std::printf("%d failures in total\n", fails);
    return fails ? 1 : 0;
}
