// Host-side mirror of the reference's node classes for the hot path, written against the C-ABI
// (include/lvi_hotpath.h) only.  The reference is C++/ROS 2; its toolchain is absent from this image,
// so these classes keep the reference's class / method names and data flow with plain structs in
// place of ROS messages.  A ROS 2 shim (host/ros2/*.cpp) converts messages to these structs.
//
//   ImageProjection::cloudHandler            imageProjection.cpp:222-237   (projectPointCloud + cloudExtraction)
//   FeatureExtraction::laserCloudInfoHandler featureExtraction.cpp:72-85
//   MapOptimization::{extractCloud, downsampleCurrentScan, scan2MapOptimization}  mapOptimization.cpp:931-999, 1315-1375
//   FeatureTracker::{readImage, setMask, addPoints}   feature_tracker.cpp:36-207
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <map>
#include <vector>

#include "../../include/lvi_hotpath.h"

namespace lvi_host {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& where) : std::runtime_error(where + ": status " + std::to_string(c) + " (" + lvi_last_error() + ")"), code(c) {}
};
inline int32_t check(int32_t st, const char* where) { if (st < 0) throw Error(st, where); return st; }

// lidar_odometry/msg/CloudInfo.msg:2-32 as a plain struct (PointCloud2 payloads as lvi_pt arrays)
struct CloudInfo {
    double stamp = 0;
    std::vector<int32_t> start_ring_index, end_ring_index, point_col_ind;
    std::vector<float> point_range;
    int64_t imu_available = 0, odom_available = 0;
    float imu_roll_init = 0, imu_pitch_init = 0, imu_yaw_init = 0;
    float initial_guess_x = 0, initial_guess_y = 0, initial_guess_z = 0, initial_guess_roll = 0, initial_guess_pitch = 0, initial_guess_yaw = 0;
    int64_t odom_reset_id = 0;
    std::vector<lvi_pt> cloud_deskewed, cloud_corner, cloud_surface;
};

class LidarHandle {
public:
    LidarHandle(const lvi_lidar_params& p, int device) : P(p) { check(lvi_lidar_create(&p, device, &h_), "lvi_lidar_create"); }
    ~LidarHandle() { lvi_lidar_destroy(h_); }
    LidarHandle(const LidarHandle&) = delete;
    LidarHandle& operator=(const LidarHandle&) = delete;
    lvi_lidar* get() const { return h_; }
    lvi_lidar_params P;
private:
    lvi_lidar* h_ = nullptr;
};

// ---------------------------------------------------------------------------------------------- ImageProjection
class ImageProjection {
public:
    explicit ImageProjection(LidarHandle& h) : h_(h) {}
    // imuDeskewInfo (imageProjection.cpp:354-410) on the host: the node keeps its IMU queue; call this with the
    // queue entries [time, angular velocity] that cover the scan, exactly as the reference loop walks them.
    // Returns false (deskew off) when fewer than two entries fall into the window, as `imuPointerCur <= 0` does.
    bool imuDeskewInfo(const double* imu_time, const double* ang_x, const double* ang_y, const double* ang_z, int n_imu,
                       double timeScanCur, double timeScanEnd)
    {
        imuTime_.clear(); rotX_.clear(); rotY_.clear(); rotZ_.clear();
        for (int i = 0; i < n_imu; ++i) {
            const double t = imu_time[i];
            if (t < timeScanCur - 0.01) continue;                          // queue pop :358-364
            if (t > timeScanEnd + 0.01) break;                             // :378
            if ((int)imuTime_.size() >= LVI_DESKEW_MAX_IMU) break;
            if (imuTime_.empty()) { imuTime_.push_back(t); rotX_.push_back(0); rotY_.push_back(0); rotZ_.push_back(0); continue; }   // :381-388
            const double dt = t - imuTime_.back();                         // :396-400
            rotX_.push_back(rotX_.back() + ang_x[i] * dt); rotY_.push_back(rotY_.back() + ang_y[i] * dt); rotZ_.push_back(rotZ_.back() + ang_z[i] * dt);
            imuTime_.push_back(t);
        }
        timeScanCur_ = timeScanCur;
        imu_available_ = imuTime_.size() >= 2;                             // --imuPointerCur; if (imuPointerCur <= 0) return; (:404-408)
        return imu_available_;
    }
    void clearDeskew() { imu_available_ = false; }

    // cloudHandler for one livox CustomMsg; deskews when imuDeskewInfo() found a table for this scan
    CloudInfo cloudHandler(const lvi_livox_pt* points, int32_t point_num, double stamp)
    {
        const int cap = h_.P.N_SCAN * h_.P.Horizon_SCAN;
        CloudInfo ci;
        ci.stamp = stamp;
        ci.start_ring_index.assign(h_.P.N_SCAN, 0); ci.end_ring_index.assign(h_.P.N_SCAN, 0);
        ci.point_col_ind.assign(cap, 0); ci.point_range.assign(cap, 0.f); ci.cloud_deskewed.resize(cap);
        lvi_scan_info si{cap, 0, ci.start_ring_index.data(), ci.end_ring_index.data(), ci.point_col_ind.data(), ci.point_range.data(), ci.cloud_deskewed.data()};
        if (imu_available_) {
            lvi_deskew_info dk{1, (int32_t)imuTime_.size() - 1, timeScanCur_, imuTime_.data(), rotX_.data(), rotY_.data(), rotZ_.data()};
            check(lvi_organize_scan_deskew(h_.get(), points, point_num, &dk, &si), "lvi_organize_scan_deskew");
        } else {
            check(lvi_organize_scan(h_.get(), points, point_num, &si), "lvi_organize_scan");
        }
        ci.cloud_deskewed.resize(si.n);          // point_col_ind / point_range keep their full size as in allocateMemory (:161-162)
        return ci;
    }
private:
    LidarHandle& h_;
    bool imu_available_ = false;
    double timeScanCur_ = 0.0;
    std::vector<double> imuTime_, rotX_, rotY_, rotZ_;
};

// ---------------------------------------------------------------------------------------------- FeatureExtraction
class FeatureExtraction {
public:
    explicit FeatureExtraction(LidarHandle& h) : h_(h) {}
    void laserCloudInfoHandler(CloudInfo& cloudInfo)
    {
        const int n = (int)cloudInfo.cloud_deskewed.size();
        lvi_scan_info si{n, n, cloudInfo.start_ring_index.data(), cloudInfo.end_ring_index.data(), cloudInfo.point_col_ind.data(),
                         cloudInfo.point_range.data(), cloudInfo.cloud_deskewed.data()};
        cloudInfo.cloud_corner.resize(std::max(n, 1)); cloudInfo.cloud_surface.resize(std::max(n, 1));
        lvi_cloud c{n, 0, cloudInfo.cloud_corner.data()}, s{n, 0, cloudInfo.cloud_surface.data()};
        check(lvi_extract_features(h_.get(), &si, &c, &s), "lvi_extract_features");
        cloudInfo.cloud_corner.resize(c.n); cloudInfo.cloud_surface.resize(s.n);
        freeCloudInfoMemory(cloudInfo);           // publishFeatureCloud :255-264
    }
    static void freeCloudInfoMemory(CloudInfo& ci)
    {
        ci.start_ring_index.clear(); ci.end_ring_index.clear(); ci.point_col_ind.clear(); ci.point_range.clear();
    }
private:
    LidarHandle& h_;
};

// ---------------------------------------------------------------------------------------------- mapOptimization (scan-to-map half)
class MapOptimization {
public:
    explicit MapOptimization(LidarHandle& h) : h_(h) { for (float& v : transformTobeMapped) v = 0.f; }
    float transformTobeMapped[6];
    bool isDegenerate = false;
    int laserCloudCornerLastDSNum = 0, laserCloudSurfLastDSNum = 0;
    lvi_icp_result last{};

    // extractCloud's downsample of laserCloud{Corner,Surf}FromMap + the kd-tree rebuild of scan2MapOptimization
    void extractCloud(const std::vector<lvi_pt>& laserCloudCornerFromMap, const std::vector<lvi_pt>& laserCloudSurfFromMap)
    {
        check(lvi_map_set(h_.get(), laserCloudCornerFromMap.data(), (int32_t)laserCloudCornerFromMap.size(),
                          laserCloudSurfFromMap.data(), (int32_t)laserCloudSurfFromMap.size()), "lvi_map_set");
        haveMap_ = true;
    }
    // f-4: saveKeyFramesAndFactor's two push_backs (:1594-1599) with the clouds staying on the device …
    int saveKeyFrame()
    {
        int32_t idx = -1;
        check(lvi_keyframe_add_current(h_.get(), transformTobeMapped, &idx), "lvi_keyframe_add_current");
        return idx;
    }
    int saveKeyFrame(const std::vector<lvi_pt>& cornerDS, const std::vector<lvi_pt>& surfDS, const float pose[6])
    {
        int32_t idx = -1;
        check(lvi_keyframe_add(h_.get(), cornerDS.data(), (int32_t)cornerDS.size(), surfDS.data(), (int32_t)surfDS.size(), pose, &idx), "lvi_keyframe_add");
        return idx;
    }
    // … and extractCloud (:931-965) for the key indices extractNearby selected (cloudToExtract[i].intensity)
    void extractCloud(const std::vector<int32_t>& keyInds)
    {
        check(lvi_map_assemble(h_.get(), keyInds.data(), (int32_t)keyInds.size()), "lvi_map_assemble");
        haveMap_ = true;
    }
    // downsampleCurrentScan + scan2MapOptimization; returns the soft status (LVI_OK, LVI_TOO_FEW_FEATURES, …)
    int32_t laserCloudInfoHandler(const CloudInfo& cloudInfo)
    {
        if (!haveMap_) return LVI_NO_MAP;                     // cloudKeyPoses3D->points.empty() (:1317)
        lvi_imu_hint imu{(int32_t)cloudInfo.imu_available, cloudInfo.imu_roll_init, cloudInfo.imu_pitch_init, cloudInfo.imu_yaw_init};
        const int32_t st = check(lvi_scan_to_map(h_.get(), cloudInfo.cloud_corner.data(), (int32_t)cloudInfo.cloud_corner.size(),
                                                 cloudInfo.cloud_surface.data(), (int32_t)cloudInfo.cloud_surface.size(),
                                                 &imu, transformTobeMapped, &last), "lvi_scan_to_map");
        isDegenerate = last.degenerate != 0;
        laserCloudCornerLastDSNum = last.n_corner_ds; laserCloudSurfLastDSNum = last.n_surf_ds;
        return st;
    }
private:
    LidarHandle& h_;
    bool haveMap_ = false;
};

// ---------------------------------------------------------------------------------------------- FeatureTracker
struct Point2f { float x, y; };

class TrackerHandle {
public:
    TrackerHandle(const lvi_tracker_params& p, int device) : P(p) { check(lvi_tracker_create(&p, device, &t_), "lvi_tracker_create"); }
    ~TrackerHandle() { lvi_tracker_destroy(t_); }
    TrackerHandle(const TrackerHandle&) = delete;
    TrackerHandle& operator=(const TrackerHandle&) = delete;
    lvi_tracker* get() const { return t_; }
    lvi_tracker_params P;
private:
    lvi_tracker* t_ = nullptr;
};

inline int cvRound(double v) { return (int)std::lrint(v); }

// cv::circle(img, center, radius, 0, -1): OpenCV's filled midpoint circle (imgproc/src/drawing.cpp Circle()),
// restated from the published algorithm — OpenCV is not vendored in the reference (parity unpinned).
inline void fillCircleZero(std::vector<uint8_t>& img, int w, int h, int cx, int cy, int radius)
{
    auto hline = [&](int y, int x0, int x1) {
        if (y < 0 || y >= h) return;
        x0 = std::max(x0, 0); x1 = std::min(x1, w - 1);
        for (int x = x0; x <= x1; x++) img[(size_t)y * w + x] = 0;
    };
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hline(cy - dy, cx - dx, cx + dx); hline(cy + dy, cx - dx, cx + dx);
        hline(cy - dx, cx - dy, cx + dy); hline(cy + dx, cx - dy, cx + dy);
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

class FeatureTracker {
public:
    FeatureTracker(TrackerHandle& t, int row, int col, int max_cnt, int min_dist) : t_(t), ROW(row), COL(col), MAX_CNT(max_cnt), MIN_DIST(min_dist) {}

    std::vector<Point2f> prev_pts, cur_pts, forw_pts, n_pts;
    std::vector<int> ids, track_cnt;
    std::vector<uint8_t> mask;
    bool PUB_THIS_FRAME = true;
    static int& n_id() { static int v = 0; return v; }

    bool inBorder(const Point2f& pt) const                      // feature_tracker.cpp:5-11
    {
        const int BORDER_SIZE = 1;
        const int img_x = cvRound(pt.x), img_y = cvRound(pt.y);
        return BORDER_SIZE <= img_x && img_x < COL - BORDER_SIZE && BORDER_SIZE <= img_y && img_y < ROW - BORDER_SIZE;
    }
    template <class T> static void reduceVector(std::vector<T>& v, const std::vector<uint8_t>& status)   // :13-29
    {
        int j = 0;
        for (int i = 0; i < (int)v.size(); i++) if (status[i]) v[j++] = v[i];
        v.resize(j);
    }
    void setMask()                                              // :36-69 (FISHEYE == 0)
    {
        mask.assign((size_t)ROW * COL, 255);
        std::vector<std::pair<int, std::pair<Point2f, int>>> cnt_pts_id;
        for (size_t i = 0; i < forw_pts.size(); i++) cnt_pts_id.push_back({track_cnt[i], {forw_pts[i], ids[i]}});
        std::sort(cnt_pts_id.begin(), cnt_pts_id.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        forw_pts.clear(); ids.clear(); track_cnt.clear();
        for (auto& it : cnt_pts_id) {
            const int x = cvRound(it.second.first.x), y = cvRound(it.second.first.y);   // Mat::at<uchar>(Point2f) → Point(cvRound)
            if (x < 0 || y < 0 || x >= COL || y >= ROW) continue;
            if (mask[(size_t)y * COL + x] == 255) {
                forw_pts.push_back(it.second.first); ids.push_back(it.second.second); track_cnt.push_back(it.first);
                fillCircleZero(mask, COL, ROW, x, y, MIN_DIST);
            }
        }
    }
    void addPoints()                                            // :71-79
    {
        for (auto& p : n_pts) { forw_pts.push_back(p); ids.push_back(-1); track_cnt.push_back(1); }
    }
    // EQUALIZE (feature_tracker.cpp:86-90): CLAHE(3.0, 8x8) on the device, inside push_image
    void setEqualize(bool on) { check(lvi_tracker_set_equalize(t_.get(), on ? 1 : 0, 3.0, 8, 8), "lvi_tracker_set_equalize"); }
    void setCamera(const lvi_mei_params& cam) { cam_ = cam; have_cam_ = true; }

    std::vector<Point2f> cur_un_pts, pts_velocity;
    std::map<int, Point2f> cur_un_pts_map, prev_un_pts_map;
    double cur_time = 0.0, prev_time = 0.0;

    // readImage without rejectWithF (findFundamentalMat RANSAC stays on the host).  `img` is ROW x COL,
    // 8-bit, tightly packed.
    void readImage(const uint8_t* img, double _cur_time = 0.0)
    {
        cur_time = _cur_time;
        check(lvi_tracker_push_image(t_.get(), img, COL, ROW, COL), "lvi_tracker_push_image");   // forw_img = img (:94-101)
        forw_pts.clear();
        if (!cur_pts.empty()) {
            std::vector<uint8_t> status(cur_pts.size());
            std::vector<float> err(cur_pts.size());
            forw_pts.resize(cur_pts.size());
            check(lvi_tracker_set_points(t_.get(), &cur_pts[0].x, (int32_t)cur_pts.size()), "lvi_tracker_set_points");
            check(lvi_tracker_run_lk(t_.get()), "lvi_tracker_run_lk");                            // calcOpticalFlowPyrLK (:113)
            int32_t n = 0;
            check(lvi_tracker_get_lk(t_.get(), &forw_pts[0].x, status.data(), err.data(), (int32_t)cur_pts.size(), &n), "lvi_tracker_get_lk");
            for (size_t i = 0; i < forw_pts.size(); i++) if (status[i] && !inBorder(forw_pts[i])) status[i] = 0;   // :137-139
            reduceVector(prev_pts, status); reduceVector(cur_pts, status); reduceVector(forw_pts, status);
            reduceVector(ids, status); reduceVector(track_cnt, status);
        }
        for (auto& n : track_cnt) n++;                                                            // :150-151
        if (PUB_THIS_FRAME) {
            setMask();
            const int n_max_cnt = MAX_CNT - (int)forw_pts.size();
            if (n_max_cnt > 0) {
                std::vector<Point2f> out((size_t)t_.P.max_features);
                int32_t n = 0;
                check(lvi_tracker_set_mask(t_.get(), mask.data(), COL, ROW, COL), "lvi_tracker_set_mask");
                check(lvi_tracker_run_gftt(t_.get(), n_max_cnt), "lvi_tracker_run_gftt");       // goodFeaturesToTrack (:166)
                check(lvi_tracker_get_gftt(t_.get(), &out[0].x, (int32_t)out.size(), &n), "lvi_tracker_get_gftt");
                n_pts.assign(out.begin(), out.begin() + n);
            } else {
                n_pts.clear();
            }
            addPoints();
        }
        prev_pts = cur_pts;                                                                       // :200-204
        cur_pts = forw_pts;
        if (have_cam_) undistortedPoints();                                                       // :205
        prev_time = cur_time;
    }
    void undistortedPoints()                                    // :298-347, liftProjective on the device (f-3)
    {
        cur_un_pts.assign(cur_pts.size(), Point2f{0.f, 0.f});
        cur_un_pts_map.clear();
        if (!cur_pts.empty())
            check(lvi_undistort_points(t_.get(), &cam_, &cur_pts[0].x, (int32_t)cur_pts.size(), &cur_un_pts[0].x), "lvi_undistort_points");
        for (size_t i = 0; i < cur_pts.size(); i++) cur_un_pts_map.insert({ids[i], cur_un_pts[i]});
        pts_velocity.clear();
        if (!prev_un_pts_map.empty()) {
            const double dt = cur_time - prev_time;
            for (size_t i = 0; i < cur_un_pts.size(); i++) {
                Point2f v{0.f, 0.f};
                if (ids[i] != -1) {
                    auto it = prev_un_pts_map.find(ids[i]);
                    if (it != prev_un_pts_map.end()) {
                        v.x = (float)((double)(cur_un_pts[i].x - it->second.x) / dt);
                        v.y = (float)((double)(cur_un_pts[i].y - it->second.y) / dt);
                    }
                }
                pts_velocity.push_back(v);
            }
        } else {
            pts_velocity.assign(cur_pts.size(), Point2f{0.f, 0.f});
        }
        prev_un_pts_map = cur_un_pts_map;
    }
    bool updateID(unsigned int i)                               // :244-254
    {
        if (i < ids.size()) { if (ids[i] == -1) ids[i] = n_id()++; return true; }
        return false;
    }
private:
    TrackerHandle& t_;
    int ROW, COL, MAX_CNT, MIN_DIST;
    lvi_mei_params cam_{}; bool have_cam_ = false;
};

}  // namespace lvi_host
