// ROS-free replay harness over the host mirror (host/lvi_host.hpp): the three lidar_odometry
// processes (imageProjection → featureExtraction → mapOptimization) and the feature_tracker callback as
// plain function calls.  Links against any library that exports the C-ABI (liblvi_hip.so in deployment;
// the CPU tests link the oracle to exercise this host code without a GPU).
//
//   replay_main lidar <Horizon_SCAN> <scan.bin> <n_raw> <map_corner.bin> <nc> <map_surf.bin> <ns> g0 g1 g2 g3 g4 g5
//   replay_main track <w> <h> <img0.bin> <img1.bin> <max_cnt> <min_dist>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "lvi_host.hpp"

template <class T>
static std::vector<T> read_file(const char* path, size_t n)
{
    std::vector<T> v(n);
    std::ifstream f(path, std::ios::binary);
    if (!f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)))) { fprintf(stderr, "short read: %s\n", path); exit(2); }
    return v;
}

int main(int argc, char** argv)
{
    using namespace lvi_host;
    try {
        if (argc >= 15 && !strcmp(argv[1], "lidar")) {
            lvi_lidar_params P; lvi_lidar_params_default(&P);
            P.Horizon_SCAN = atoi(argv[2]);
            const int n_raw = atoi(argv[4]), nc = atoi(argv[6]), ns = atoi(argv[8]);
            P.max_raw_points = n_raw + 16; P.max_map_points = std::max(nc, ns) + 16;
            LidarHandle h(P, 0);
            auto scan = read_file<lvi_livox_pt>(argv[3], n_raw);
            auto mc = read_file<lvi_pt>(argv[5], nc);
            auto ms = read_file<lvi_pt>(argv[7], ns);
            ImageProjection ip(h); FeatureExtraction fe(h); MapOptimization mo(h);
            CloudInfo ci = ip.cloudHandler(scan.data(), n_raw, 0.0);
            const int n = (int)ci.cloud_deskewed.size();
            fe.laserCloudInfoHandler(ci);
            mo.extractCloud(mc, ms);
            for (int k = 0; k < 6; k++) mo.transformTobeMapped[k] = (float)atof(argv[9 + k]);
            const int st = mo.laserCloudInfoHandler(ci);
            printf("backend %s\nn %d corner %zu surf %zu cornerDS %d surfDS %d\nstatus %d iters %d degenerate %d\npose",
                   lvi_backend(), n, ci.cloud_corner.size(), ci.cloud_surface.size(), mo.laserCloudCornerLastDSNum, mo.laserCloudSurfLastDSNum,
                   st, mo.last.iters, (int)mo.isDegenerate);
            for (int k = 0; k < 6; k++) printf(" %.9g", mo.transformTobeMapped[k]);
            printf("\n");
            return 0;
        }
        if (argc >= 8 && !strcmp(argv[1], "track")) {
            const int w = atoi(argv[2]), hgt = atoi(argv[3]);
            lvi_tracker_params P; lvi_tracker_params_default(&P);
            P.max_width = w; P.max_height = hgt; P.max_cnt = atoi(argv[6]); P.min_dist = atof(argv[7]);
            TrackerHandle t(P, 0);
            FeatureTracker ft(t, hgt, w, P.max_cnt, (int)P.min_dist);
            auto a = read_file<uint8_t>(argv[4], (size_t)w * hgt);
            auto b = read_file<uint8_t>(argv[5], (size_t)w * hgt);
            ft.readImage(a.data());
            for (unsigned i = 0; ft.updateID(i); i++) {}
            const size_t n0 = ft.cur_pts.size();
            ft.readImage(b.data());
            for (unsigned i = 0; ft.updateID(i); i++) {}
            size_t tracked = 0;
            for (int c : ft.track_cnt) tracked += c > 1;
            printf("backend %s\nfirst %zu second %zu tracked %zu\n", lvi_backend(), n0, ft.cur_pts.size(), tracked);
            for (size_t i = 0; i < ft.cur_pts.size(); i++) printf("%d %d %.9g %.9g\n", ft.ids[i], ft.track_cnt[i], ft.cur_pts[i].x, ft.cur_pts[i].y);
            return 0;
        }
        fprintf(stderr, "usage: see the header of replay_main.cpp\n");
        return 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
}
