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
            if (const char* e = getenv("LVI_NUMBER_OF_CORES")) P.numberOfCores = atoi(e);     // OpenMP threads of the CPU library (tests)
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
        if (argc >= 7 && !strcmp(argv[1], "extras")) {
            // replay_main extras <Horizon_SCAN> <scan.bin> <n_raw> <w> <h> <img.bin>
            // f-1 deskew through ImageProjection::imuDeskewInfo, f-4 keyframes + map assembly + matching against it,
            // f-2 equalised readImage, f-3 undistortedPoints: the rows next to the path, through the host mirror
            lvi_lidar_params P; lvi_lidar_params_default(&P);
            if (const char* e = getenv("LVI_NUMBER_OF_CORES")) P.numberOfCores = atoi(e);     // OpenMP threads of the CPU library (tests)
            P.Horizon_SCAN = atoi(argv[2]);
            const int n_raw = atoi(argv[4]);
            P.max_raw_points = n_raw + 16; P.max_map_points = 1 << 20;
            LidarHandle h(P, 0);
            auto scan = read_file<lvi_livox_pt>(argv[3], n_raw);
            ImageProjection ip(h); FeatureExtraction fe(h); MapOptimization mo(h);
            // a 200 Hz IMU turning at 0.5 rad/s about z, scan stamped at t = 100 s
            std::vector<double> t, wx, wy, wz;
            for (int i = 0; i < 40; i++) { t.push_back(99.99 + 0.005 * i); wx.push_back(0.0); wy.push_back(0.0); wz.push_back(0.5); }
            const bool dk = ip.imuDeskewInfo(t.data(), wx.data(), wy.data(), wz.data(), (int)t.size(), 100.0, 100.1);
            CloudInfo ci = ip.cloudHandler(scan.data(), n_raw, 100.0);
            double sx = 0, sy = 0;
            for (const lvi_pt& p : ci.cloud_deskewed) { sx += p.x; sy += p.y; }
            printf("backend %s\ndeskew %d n %zu sum %.6f %.6f\n", lvi_backend(), (int)dk, ci.cloud_deskewed.size(), sx, sy);
            ip.clearDeskew();
            ci = ip.cloudHandler(scan.data(), n_raw, 100.0);
            fe.laserCloudInfoHandler(ci);
            // the scan becomes keyframe 0 (identity pose) and keyframe 1 (shifted by 0.05 m): the fused map is what the
            // same scan is then matched against, starting 0.1 m off
            const float pose0[6] = {0, 0, 0, 0, 0, 0}, pose1[6] = {0, 0, 0, 0.05f, 0, 0};
            const int k0 = mo.saveKeyFrame(ci.cloud_corner, ci.cloud_surface, pose0);
            const int k1 = mo.saveKeyFrame(ci.cloud_corner, ci.cloud_surface, pose1);
            mo.extractCloud(std::vector<int32_t>{k0, k1});
            mo.transformTobeMapped[3] = 0.1f;
            const int st = mo.laserCloudInfoHandler(ci);
            printf("keys %d %d status %d iters %d pose", k0, k1, st, mo.last.iters);
            for (int k = 0; k < 6; k++) printf(" %.9g", mo.transformTobeMapped[k]);
            printf("\n");
            const int w = atoi(argv[5]), hgt = atoi(argv[6]);
            lvi_tracker_params TP; lvi_tracker_params_default(&TP);
            TP.max_width = w; TP.max_height = hgt; TP.max_cnt = 40; TP.min_dist = 12;
            TrackerHandle th(TP, 0);
            FeatureTracker ft(th, hgt, w, TP.max_cnt, (int)TP.min_dist);
            ft.setEqualize(true);
            const lvi_mei_params cam{1.9926618269451453, -0.0399258932468764, 0.15160828121223818, 0.00017756967825777937, -0.0011531239076798612,
                                     669.8940458885896, 669.1450614220616, 0.5 * w, 0.5 * hgt};
            ft.setCamera(cam);
            auto img = read_file<uint8_t>(argv[7], (size_t)w * hgt);
            ft.readImage(img.data(), 1.0);
            for (unsigned i = 0; ft.updateID(i); i++) {}
            ft.readImage(img.data(), 1.1);
            printf("features %zu un %zu vel %zu\n", ft.cur_pts.size(), ft.cur_un_pts.size(), ft.pts_velocity.size());
            for (size_t i = 0; i < ft.cur_pts.size(); i++)
                printf("%.9g %.9g %.9g %.9g %.9g %.9g\n", ft.cur_pts[i].x, ft.cur_pts[i].y, ft.cur_un_pts[i].x, ft.cur_un_pts[i].y, ft.pts_velocity[i].x, ft.pts_velocity[i].y);
            return 0;
        }
        fprintf(stderr, "usage: see the header of replay_main.cpp\n");
        return 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
}
