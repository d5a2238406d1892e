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
#include <functional>
#include <set>
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

// ---------------------------------------------------------------------------------------------- mapOptimization (caller loop)
// The part of mapOptimization around scan matching that decides WHICH keyframes form the local map and WHEN a scan
// becomes a keyframe: updateInitialGuess (:806-877), extractNearby (:894-929), extractCloud's
// key selection (:931-957), saveFrame (:1387-1412), the key-pose push of saveKeyFramesAndFactor (:1575-1599).
// Key POSES are a few thousand points: host work.  Key CLOUDS never leave the device (lvi_keyframe_*, lvi_map_*).
// iSAM2 / loop closure / GPS are out of scope: with only the prior and odometry factors the newest
// estimate of iSAM2 is the scan-matching result, which is what is pushed here ("odometry chain").
struct Affine3f { float m[3][4]; };            // Eigen::Affine3f: linear part | translation
inline Affine3f getTransformation(float x, float y, float z, float roll, float pitch, float yaw)      // pcl::getTransformation
{
    const float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll), F = std::sin(roll), DE = D * E, DF = D * F;
    return Affine3f{{{A * C, A * DF - B * E, B * F + A * DE, x}, {B * C, A * E + B * DF, B * DE - A * F, y}, {-D, C * F, C * E, z}}};
}
inline Affine3f affineInverse(const Affine3f& t)                      // Eigen Transform::inverse(Affine): cofactor inverse of the 3x3, -inv * translation
{
    const float (*R)[4] = t.m;
    auto cof = [&](int i, int j) { return R[(i + 1) % 3][(j + 1) % 3] * R[(i + 2) % 3][(j + 2) % 3] - R[(i + 1) % 3][(j + 2) % 3] * R[(i + 2) % 3][(j + 1) % 3]; };
    const float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    const float invdet = 1.0f / ((c0 * R[0][0] + c1 * R[1][0]) + c2 * R[2][0]);
    Affine3f o;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.m[i][j] = cof(j, i) * invdet;
    for (int i = 0; i < 3; i++) o.m[i][3] = -((o.m[i][0] * R[0][3] + o.m[i][1] * R[1][3]) + o.m[i][2] * R[2][3]);
    return o;
}
inline Affine3f affineMul(const Affine3f& a, const Affine3f& b)
{
    Affine3f o;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) o.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
        o.m[i][3] = ((a.m[i][0] * b.m[0][3] + a.m[i][1] * b.m[1][3]) + a.m[i][2] * b.m[2][3]) + a.m[i][3];
    }
    return o;
}
inline void getTranslationAndEulerAngles(const Affine3f& t, float& x, float& y, float& z, float& roll, float& pitch, float& yaw)   // pcl (App. A.3)
{
    x = t.m[0][3]; y = t.m[1][3]; z = t.m[2][3];
    roll = std::atan2(t.m[2][1], t.m[2][2]); pitch = std::asin(-t.m[2][0]); yaw = std::atan2(t.m[1][0], t.m[0][0]);
}

struct PointTypePose { float x, y, z, intensity, roll, pitch, yaw; double time; };     // utility.h:71-78

struct MapCallerParams {                       // params_lidar.yaml:17,60-66 (utility.h:176,278-288)
    bool   useImuHeadingInitialization = false;
    double mappingProcessInterval = 0.15;
    float  surroundingkeyframeAddingDistThreshold = 1.0f;
    float  surroundingkeyframeAddingAngleThreshold = 0.2f;
    float  surroundingKeyframeDensity = 2.0f;
    float  surroundingKeyframeSearchRadius = 50.0f;
    bool   sensorIsLivox = true;               // sensor == SensorType::LIVOX (:1392-1396)
    bool   incrementalMap = true;              // lvi_map_update (keyframes entering / leaving the local map) instead of lvi_map_assemble; same bits
};

class MapOptimizationNode {
public:
    MapOptimizationNode(LidarHandle& h, const MapCallerParams& p = MapCallerParams()) : P(p), h_(h) { for (float& v : transformTobeMapped) v = 0.f; }
    MapCallerParams P;
    float transformTobeMapped[6];
    std::vector<lvi_pt> cloudKeyPoses3D;       // x, y, z, intensity = index
    std::vector<PointTypePose> cloudKeyPoses6D;
    std::vector<int32_t> lastKeys;             // key indices of the last extractCloud, in fuse order
    lvi_icp_result last{};
    int32_t lastStatus = LVI_NO_MAP;
    bool lastSavedKeyFrame = false, lastProcessed = false;

    // laserCloudInfoHandler (:298-333) with the feature clouds of the message.  Returns false when the
    // mappingProcessInterval gate dropped the scan.
    bool laserCloudInfoHandler(const CloudInfo& cloudInfo)
    {
        timeLaserInfoCur = cloudInfo.stamp;
        lastProcessed = false;
        if (!(timeLaserInfoCur - timeLastProcessing >= P.mappingProcessInterval)) return false;
        timeLastProcessing = timeLaserInfoCur;
        lastProcessed = true;
        updateInitialGuess(cloudInfo);
        extractSurroundingKeyFrames();
        // downsampleCurrentScan + scan2MapOptimization (:987-999, 1315-1343): one C-ABI call; without key poses it only downsamples
        lvi_imu_hint imu{(int32_t)cloudInfo.imu_available, cloudInfo.imu_roll_init, cloudInfo.imu_pitch_init, cloudInfo.imu_yaw_init};
        lastStatus = check(lvi_scan_to_map(h_.get(), cloudInfo.cloud_corner.data(), (int32_t)cloudInfo.cloud_corner.size(),
                                           cloudInfo.cloud_surface.data(), (int32_t)cloudInfo.cloud_surface.size(), &imu, transformTobeMapped, &last),
                           "lvi_scan_to_map");
        saveKeyFramesAndFactor();
        return true;
    }
    // the same for a scan whose features already sit on the device (replay harness: lvi_scan_upload* / organize / extract ran)
    bool processResidentScan(double stamp, const lvi_imu_hint& imu)
    {
        CloudInfo ci; ci.stamp = stamp; ci.imu_available = imu.imu_available;
        ci.imu_roll_init = imu.imu_roll_init; ci.imu_pitch_init = imu.imu_pitch_init; ci.imu_yaw_init = imu.imu_yaw_init;
        timeLaserInfoCur = stamp;
        lastProcessed = false;
        if (!(timeLaserInfoCur - timeLastProcessing >= P.mappingProcessInterval)) return false;
        timeLastProcessing = timeLaserInfoCur;
        lastProcessed = true;
        updateInitialGuess(ci);
        // downsampleCurrentScan (:987-999) is enqueued BEFORE extractSurroundingKeyFrames: the two do not depend on each other, and
        // the scan's grids then run on the main stream while the local map is updated on the handle's second stream
        check(lvi_scan_downsample(h_.get()), "lvi_scan_downsample");
        extractSurroundingKeyFrames();
        lastStatus = check(lvi_scan_match(h_.get(), &imu, transformTobeMapped, &last), "lvi_scan_match");
        saveKeyFramesAndFactor();
        return true;
    }

    void updateInitialGuess(const CloudInfo& cloudInfo)                                   // :806-877
    {
        if (cloudKeyPoses3D.empty()) {
            transformTobeMapped[0] = cloudInfo.imu_roll_init; transformTobeMapped[1] = cloudInfo.imu_pitch_init; transformTobeMapped[2] = cloudInfo.imu_yaw_init;
            if (!P.useImuHeadingInitialization) transformTobeMapped[2] = 0;
            lastImuTransformation = getTransformation(0, 0, 0, cloudInfo.imu_roll_init, cloudInfo.imu_pitch_init, cloudInfo.imu_yaw_init);
            return;
        }
        if (cloudInfo.odom_available && cloudInfo.odom_reset_id == odom_reset_id) {       // VINS odometry guess (:825-851)
            const Affine3f transBack = getTransformation(cloudInfo.initial_guess_x, cloudInfo.initial_guess_y, cloudInfo.initial_guess_z,
                                                         cloudInfo.initial_guess_roll, cloudInfo.initial_guess_pitch, cloudInfo.initial_guess_yaw);
            if (!lastVinsTransAvailable) {
                lastVinsTransformation = transBack; lastVinsTransAvailable = true;
            } else {
                applyIncrement(affineMul(affineInverse(lastVinsTransformation), transBack));
                lastVinsTransformation = transBack;
                lastImuTransformation = getTransformation(0, 0, 0, cloudInfo.imu_roll_init, cloudInfo.imu_pitch_init, cloudInfo.imu_yaw_init);
                return;
            }
        } else {
            lastVinsTransAvailable = false; odom_reset_id = cloudInfo.odom_reset_id;
        }
        if (cloudInfo.imu_available) {                                                    // IMU rotation increment (:859-871)
            const Affine3f transBack = getTransformation(0, 0, 0, cloudInfo.imu_roll_init, cloudInfo.imu_pitch_init, cloudInfo.imu_yaw_init);
            applyIncrement(affineMul(affineInverse(lastImuTransformation), transBack));
            lastImuTransformation = transBack;
        }
    }

    void extractSurroundingKeyFrames() { if (!cloudKeyPoses3D.empty()) extractNearby(); }   // :968-981

    void extractNearby()                                                                  // :894-929
    {
        const lvi_pt back = cloudKeyPoses3D.back();
        const double radius = (double)P.surroundingKeyframeSearchRadius;
        // kdtree radiusSearch (sorted by distance; brute force over the key poses, ties by index)
        std::vector<std::pair<float, int>> hits;
        for (int i = 0; i < (int)cloudKeyPoses3D.size(); i++) {
            const float d = sqDist(cloudKeyPoses3D[i], back);
            if ((double)d <= radius * radius) hits.push_back({d, i});
        }
        std::stable_sort(hits.begin(), hits.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        std::vector<lvi_pt> surroundingKeyPoses;
        for (auto& hpair : hits) surroundingKeyPoses.push_back(cloudKeyPoses3D[hpair.second]);
        // downSizeFilterSurroundingKeyPoses: the library's VoxelGrid, leaf surroundingKeyframeDensity
        std::vector<lvi_pt> surroundingKeyPosesDS(std::max<size_t>(surroundingKeyPoses.size(), 1));
        int32_t nds = 0;
        check(lvi_voxel_downsample(h_.get(), surroundingKeyPoses.data(), (int32_t)surroundingKeyPoses.size(), P.surroundingKeyframeDensity,
                                   surroundingKeyPosesDS.data(), (int32_t)surroundingKeyPosesDS.size(), &nds), "lvi_voxel_downsample(key poses)");
        surroundingKeyPosesDS.resize(nds);
        for (auto& pt : surroundingKeyPosesDS) {                                          // nearestKSearch(pt, 1) → the key's index
            int best = 0; float bd = sqDist(cloudKeyPoses3D[0], pt);
            for (int i = 1; i < (int)cloudKeyPoses3D.size(); i++) { const float d = sqDist(cloudKeyPoses3D[i], pt); if (d < bd) { bd = d; best = i; } }
            pt.intensity = cloudKeyPoses3D[best].intensity;
        }
        for (int i = (int)cloudKeyPoses3D.size() - 1; i >= 0; --i) {                      // the latest key frames (robot rotating in place)
            if (timeLaserInfoCur - cloudKeyPoses6D[i].time < 10.0) surroundingKeyPosesDS.push_back(cloudKeyPoses3D[i]);
            else break;
        }
        extractCloud(surroundingKeyPosesDS);
    }

    void extractCloud(const std::vector<lvi_pt>& cloudToExtract)                          // :931-965
    {
        lastKeys.clear();
        for (const lvi_pt& pt : cloudToExtract) {
            if (std::sqrt(sqDist(pt, cloudKeyPoses3D.back())) > P.surroundingKeyframeSearchRadius) continue;     // pointDistance (utility.h:408-411)
            lastKeys.push_back((int32_t)pt.intensity);
        }
        if (P.incrementalMap) check(lvi_map_update(h_.get(), lastKeys.data(), (int32_t)lastKeys.size()), "lvi_map_update");
        else check(lvi_map_assemble(h_.get(), lastKeys.data(), (int32_t)lastKeys.size()), "lvi_map_assemble");
    }

    // a keyframe from an earlier session (map loaded at start-up): its DS clouds go into the device store, its pose into
    // cloudKeyPoses3D / 6D, exactly what saveKeyFramesAndFactor leaves behind for a scan it kept
    int seedKeyFrame(const lvi_pt* cornerDS, int32_t nc, const lvi_pt* surfDS, int32_t ns, const float pose[6], double time)
    {
        int32_t idx = -1;
        check(lvi_keyframe_add(h_.get(), cornerDS, nc, surfDS, ns, pose, &idx), "lvi_keyframe_add");
        if (idx != (int32_t)cloudKeyPoses3D.size()) throw Error(LVI_ERR_STATE, "keyframe store out of step with the key poses");
        cloudKeyPoses3D.push_back(lvi_pt{pose[3], pose[4], pose[5], (float)idx});
        cloudKeyPoses6D.push_back(PointTypePose{pose[3], pose[4], pose[5], (float)idx, pose[0], pose[1], pose[2], time});
        for (int k = 0; k < 6; k++) transformTobeMapped[k] = pose[k];
        return idx;
    }

    bool saveFrame() const                                                                // :1387-1412
    {
        if (cloudKeyPoses3D.empty()) return true;
        if (P.sensorIsLivox && timeLaserInfoCur - cloudKeyPoses6D.back().time > 1.0) return true;
        const PointTypePose& b = cloudKeyPoses6D.back();
        const Affine3f transStart = getTransformation(b.x, b.y, b.z, b.roll, b.pitch, b.yaw);      // pclPointToAffine3f
        const Affine3f transFinal = getTransformation(transformTobeMapped[3], transformTobeMapped[4], transformTobeMapped[5],
                                                      transformTobeMapped[0], transformTobeMapped[1], transformTobeMapped[2]);
        float x, y, z, roll, pitch, yaw;
        getTranslationAndEulerAngles(affineMul(affineInverse(transStart), transFinal), x, y, z, roll, pitch, yaw);
        if (std::abs(roll) < P.surroundingkeyframeAddingAngleThreshold && std::abs(pitch) < P.surroundingkeyframeAddingAngleThreshold &&
            std::abs(yaw) < P.surroundingkeyframeAddingAngleThreshold && std::sqrt(x * x + y * y + z * z) < P.surroundingkeyframeAddingDistThreshold)
            return false;
        return true;
    }

    void saveKeyFramesAndFactor()                                                         // :1529-1603 without the factor graph
    {
        lastSavedKeyFrame = false;
        if (!saveFrame()) return;
        lvi_pt thisPose3D{transformTobeMapped[3], transformTobeMapped[4], transformTobeMapped[5], (float)cloudKeyPoses3D.size()};
        PointTypePose thisPose6D{thisPose3D.x, thisPose3D.y, thisPose3D.z, thisPose3D.intensity,
                                 transformTobeMapped[0], transformTobeMapped[1], transformTobeMapped[2], timeLaserInfoCur};
        int32_t idx = -1;                                                                 // cornerCloudKeyFrames / surfCloudKeyFrames push_back: device to device
        check(lvi_keyframe_add_current(h_.get(), transformTobeMapped, &idx), "lvi_keyframe_add_current");
        if (idx != (int32_t)cloudKeyPoses3D.size()) throw Error(LVI_ERR_STATE, "keyframe store out of step with the key poses");
        cloudKeyPoses3D.push_back(thisPose3D); cloudKeyPoses6D.push_back(thisPose6D);
        lastSavedKeyFrame = true;
    }
private:
    static float sqDist(const lvi_pt& a, const lvi_pt& b)
    {
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        return (dx * dx + dy * dy) + dz * dz;
    }
    void applyIncrement(const Affine3f& transIncre)
    {
        const Affine3f transTobe = getTransformation(transformTobeMapped[3], transformTobeMapped[4], transformTobeMapped[5],
                                                     transformTobeMapped[0], transformTobeMapped[1], transformTobeMapped[2]);      // trans2Affine3f
        getTranslationAndEulerAngles(affineMul(transTobe, transIncre), transformTobeMapped[3], transformTobeMapped[4], transformTobeMapped[5],
                                     transformTobeMapped[0], transformTobeMapped[1], transformTobeMapped[2]);
    }
    LidarHandle& h_;
    double timeLaserInfoCur = 0.0, timeLastProcessing = -1.0;
    Affine3f lastImuTransformation{}, lastVinsTransformation{};
    bool lastVinsTransAvailable = false;
    int64_t odom_reset_id = 0;
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
    // cv::circle(mask, pt, r, 0, -1) covers pixel (x, y) <=> |y - cy| <= r and |x - cx| <= hw[|y - cy|], hw = the widest horizontal span
    // OpenCV's midpoint raster (FillCircle) draws at that row offset (fillCircleZero above draws exactly these spans)
    static std::vector<int> circleHalfWidths(int radius)
    {
        std::vector<int> hw((size_t)radius + 1, -1);
        int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
        while (dx >= dy) {
            hw[dy] = std::max(hw[dy], dx); hw[dx] = std::max(hw[dx], dy);
            dy++;
            err += plus; plus += 2;
            const int m = (err <= 0) - 1;
            err -= minus & m; dx += m; minus -= m & 2;
        }
        return hw;
    }
    // :36-69 (FISHEYE == 0).  The order by track_cnt and the walk "keep a point if the mask is still 255 there, then blank MIN_DIST
    // around it" stay on the host (150 points); "is the mask still 255 at (x, y)" is answered from the kept points' circles instead
    // of a W x H image, and the image itself is rastered on the device from the kept points (lvi_tracker_set_mask_circles): nothing
    // of its 0.6 - 0.9 MB crosses the bus.  `mask` (host image) is only materialised on request (keep_host_mask: tests).
    bool keep_host_mask = false;
    void setMask()
    {
        std::vector<std::pair<int, std::pair<Point2f, int>>> cnt_pts_id;
        for (size_t i = 0; i < forw_pts.size(); i++) cnt_pts_id.push_back({track_cnt[i], {forw_pts[i], ids[i]}});
        std::sort(cnt_pts_id.begin(), cnt_pts_id.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        forw_pts.clear(); ids.clear(); track_cnt.clear();
        if (hw_.size() != (size_t)MIN_DIST + 1) hw_ = circleHalfWidths(MIN_DIST);
        std::vector<std::pair<int, int>> kept_px;
        for (auto& it : cnt_pts_id) {
            const int x = cvRound(it.second.first.x), y = cvRound(it.second.first.y);   // Mat::at<uchar>(Point2f) → Point(cvRound)
            if (x < 0 || y < 0 || x >= COL || y >= ROW) continue;
            bool free_px = true;
            for (const auto& c : kept_px) {
                const int ady = std::abs(y - c.second), adx = std::abs(x - c.first);
                if (ady <= MIN_DIST && adx <= hw_[ady]) { free_px = false; break; }
            }
            if (free_px) {
                forw_pts.push_back(it.second.first); ids.push_back(it.second.second); track_cnt.push_back(it.first);
                kept_px.push_back({x, y});
            }
        }
        if (keep_host_mask) {
            mask.assign((size_t)ROW * COL, 255);
            for (const auto& c : kept_px) fillCircleZero(mask, COL, ROW, c.first, c.second, MIN_DIST);
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

    // cv::findFundamentalMat(un_cur_pts, un_forw_pts, cv::FM_RANSAC, F_THRESHOLD, 0.99, status) of rejectWithF
    // (feature_tracker.cpp:209-242).  The RANSAC itself is host-side OpenCV in the reference and stays with the node
    // (not a kernel); the node installs it here.  Without a hook — or without a camera — readImage
    // skips the call and says so in rejectWithF_skipped, so a caller comparing against the reference knows that the
    // point set entering setMask may be larger than the reference's.
    using FundamentalMatFn = std::function<void(const std::vector<Point2f>& un_cur_pts, const std::vector<Point2f>& un_forw_pts,
                                                double F_THRESHOLD, std::vector<uint8_t>& status)>;
    FundamentalMatFn findFundamentalMat;
    double F_THRESHOLD = 1.0;                                   // params_camera.yaml F_threshold
    int FOCAL_LENGTH = 460;                                     // parameters.cpp:101
    int rejectWithF_skipped = 0;                                // PUB frames on which >= 8 points were tracked but no hook was installed

    // CataCamera::liftProjective (CataCamera.cc:556-626, distortion :766-783) in double on the host: rejectWithF needs the
    // double ray before it is scaled by FOCAL_LENGTH (the device version, lvi_undistort_points, returns f32 x/z, y/z)
    void liftProjective(double px, double py, double P[3]) const
    {
        const double inv_K11 = 1.0 / cam_.gamma1, inv_K13 = -cam_.u0 / cam_.gamma1, inv_K22 = 1.0 / cam_.gamma2, inv_K23 = -cam_.v0 / cam_.gamma2;
        const double mx_d = inv_K11 * px + inv_K13, my_d = inv_K22 * py + inv_K23;
        auto distortion = [&](double ux, double uy, double& dx, double& dy) {
            const double mx2_u = ux * ux, my2_u = uy * uy, mxy_u = ux * uy, rho2_u = mx2_u + my2_u;
            const double rad_dist_u = cam_.k1 * rho2_u + cam_.k2 * rho2_u * rho2_u;
            dx = ux * rad_dist_u + 2.0 * cam_.p1 * mxy_u + cam_.p2 * (rho2_u + 2.0 * mx2_u);
            dy = uy * rad_dist_u + 2.0 * cam_.p2 * mxy_u + cam_.p1 * (rho2_u + 2.0 * my2_u);
        };
        double dx, dy;
        distortion(mx_d, my_d, dx, dy);
        double mx_u = mx_d - dx, my_u = my_d - dy;
        for (int i = 1; i < 8; ++i) { distortion(mx_u, my_u, dx, dy); mx_u = mx_d - dx; my_u = my_d - dy; }
        const double xi = cam_.xi;
        P[0] = mx_u; P[1] = my_u;
        if (xi == 1.0) P[2] = (1.0 - mx_u * mx_u - my_u * my_u) / 2.0;
        else { const double rho2_d = mx_u * mx_u + my_u * my_u; P[2] = 1.0 - xi * (rho2_d + 1.0) / (xi + std::sqrt(1.0 + (1.0 - xi * xi) * rho2_d)); }
    }
    void rejectWithF()                                          // feature_tracker.cpp:209-242
    {
        if (forw_pts.size() < 8) return;
        if (!findFundamentalMat || !have_cam_) { rejectWithF_skipped++; return; }
        std::vector<Point2f> un_cur_pts(cur_pts.size()), un_forw_pts(forw_pts.size());
        for (size_t i = 0; i < cur_pts.size(); i++) {
            double tmp_p[3];
            liftProjective(cur_pts[i].x, cur_pts[i].y, tmp_p);
            un_cur_pts[i] = Point2f{(float)(FOCAL_LENGTH * tmp_p[0] / tmp_p[2] + COL / 2.0), (float)(FOCAL_LENGTH * tmp_p[1] / tmp_p[2] + ROW / 2.0)};
            liftProjective(forw_pts[i].x, forw_pts[i].y, tmp_p);
            un_forw_pts[i] = Point2f{(float)(FOCAL_LENGTH * tmp_p[0] / tmp_p[2] + COL / 2.0), (float)(FOCAL_LENGTH * tmp_p[1] / tmp_p[2] + ROW / 2.0)};
        }
        std::vector<uint8_t> status(cur_pts.size(), 1);
        findFundamentalMat(un_cur_pts, un_forw_pts, F_THRESHOLD, status);
        reduceVector(prev_pts, status); reduceVector(cur_pts, status); reduceVector(forw_pts, status);
        reduceVector(ids, status); reduceVector(track_cnt, status);
    }

    // readImage (feature_tracker.cpp:81-207).  `img` is ROW x COL, 8-bit, tightly packed.
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
        bool gftt_asked = false;
        if (PUB_THIS_FRAME) {
            rejectWithF();                                                                        // :153
            setMask();
            const int n_max_cnt = MAX_CNT - (int)forw_pts.size();
            if (n_max_cnt > 0) {
                // the mask is rastered on the device from the kept points; goodFeaturesToTrack (:166) is enqueued behind it
                check(lvi_tracker_set_mask_circles(t_.get(), forw_pts.empty() ? nullptr : &forw_pts[0].x, (int32_t)forw_pts.size(), MIN_DIST), "lvi_tracker_set_mask_circles");
                check(lvi_tracker_run_gftt_async(t_.get(), n_max_cnt), "lvi_tracker_run_gftt_async");
                gftt_asked = true;
            }
        }
        // ONE read ends the frame: the new corners and — with a camera — the undistorted cur_pts of the next frame (= forw_pts + n_pts)
        {
            std::vector<Point2f> out((size_t)t_.P.max_features), un((size_t)t_.P.max_features);
            int32_t n = 0;
            const bool want_un = have_cam_;
            if (gftt_asked || (want_un && !forw_pts.empty()))
                check(lvi_tracker_finish_frame(t_.get(), want_un ? &cam_ : nullptr, forw_pts.empty() ? nullptr : &forw_pts[0].x, (int32_t)forw_pts.size(),
                                               &out[0].x, (int32_t)out.size(), &n, want_un ? &un[0].x : nullptr), "lvi_tracker_finish_frame");
            if (PUB_THIS_FRAME) {
                n_pts.assign(out.begin(), out.begin() + (gftt_asked ? n : 0));
                addPoints();
            }
            un_ready_.assign(un.begin(), un.begin() + (want_un ? forw_pts.size() : 0));
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
        if (un_ready_.size() == cur_pts.size()) cur_un_pts = un_ready_;                           // came back with the frame's one read (lvi_tracker_finish_frame)
        else if (!cur_pts.empty())
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
    std::vector<int> hw_;                                      // circleHalfWidths(MIN_DIST)
    std::vector<Point2f> un_ready_;                            // undistorted [forw_pts ; n_pts] of this frame, from lvi_tracker_finish_frame
};

// ---------------------------------------------------------------------------------------------- feature_tracker_node
// img_callback (feature_tracker_node.cpp:37-231) for NUM_OF_CAM == 1 without ROS: first-image / discontinuity handling,
// the frequency control that sets PUB_THIS_FRAME, updateID, and the assembly of the /vins/feature/feature message
// (sensor_msgs/PointCloud: points = (un_x, un_y, 1); channels = id, u, v, vx, vy, depth; frame_id "vins_body"; only
// features with track_cnt > 1; the first assembled message is not published).
struct Point3f { float x, y, z; };
struct FeatureMsg {
    double stamp = 0.0;
    std::string frame_id;
    std::vector<Point3f> points;
    std::vector<float> channels[6];             // id_of_point, u, v, velocity_x, velocity_y, depth (in this order, :204-224)
};

class FeatureTrackerNode {
public:
    enum Outcome { FIRST_IMAGE = 0, RESTART = 1, NOT_PUBLISHED = 2, FIRST_PUBLISH_SUPPRESSED = 3, PUBLISHED = 4 };
    static constexpr int NUM_OF_CAM = 1;
    FeatureTrackerNode(FeatureTracker& ft, int freq) : trackerData(ft), FREQ(freq == 0 ? 100 : freq) {}   // parameters.cpp:104-105

    // DepthRegister::get_depth (feature_tracker.h:116-): lidar depth association is outside the hot path; the
    // node may install it.  Default = the reference's own initial value when no depth cloud is available: -1 per feature.
    std::function<std::vector<float>(double stamp, const std::vector<Point3f>& features_2d)> get_depth;
    int restarts = 0;                           // /vins/feature/restart messages that would have been published (:56-58)

    Outcome img_callback(const uint8_t* img, double cur_img_time, FeatureMsg* feature_points)
    {
        if (first_image_flag) {                                                            // :41-47
            first_image_flag = false; first_image_time = cur_img_time; last_image_time = cur_img_time;
            return FIRST_IMAGE;
        }
        if (cur_img_time - last_image_time > 1.0 || cur_img_time < last_image_time) {     // :50-59 unstable camera stream
            first_image_flag = true; last_image_time = 0; pub_count = 1; restarts++;
            return RESTART;
        }
        last_image_time = cur_img_time;
        // frequency control (:101-112)
        bool PUB_THIS_FRAME;
        if (std::round(1.0 * pub_count / (cur_img_time - first_image_time)) <= FREQ) {
            PUB_THIS_FRAME = true;
            if (std::abs(1.0 * pub_count / (cur_img_time - first_image_time) - FREQ) < 0.01 * FREQ) { first_image_time = cur_img_time; pub_count = 0; }
        } else {
            PUB_THIS_FRAME = false;
        }
        trackerData.PUB_THIS_FRAME = PUB_THIS_FRAME;
        trackerData.readImage(img, cur_img_time);                                         // :139-140
        for (unsigned int i = 0;; i++) if (!trackerData.updateID(i)) break;               // :155-164
        if (!PUB_THIS_FRAME) return NOT_PUBLISHED;
        pub_count++;                                                                      // :169
        FeatureMsg msg;
        msg.stamp = cur_img_time; msg.frame_id = "vins_body";                             // :177-178
        std::set<int> hash_ids;
        const auto& un_pts = trackerData.cur_un_pts; const auto& cur_pts = trackerData.cur_pts;
        const auto& ids = trackerData.ids; const auto& pts_velocity = trackerData.pts_velocity;
        for (unsigned int j = 0; j < ids.size(); j++) {
            if (trackerData.track_cnt[j] > 1) {                                           // :189
                const int p_id = ids[j];
                hash_ids.insert(p_id);
                msg.points.push_back(Point3f{un_pts[j].x, un_pts[j].y, 1.f});
                msg.channels[0].push_back((float)(p_id * NUM_OF_CAM + 0));
                msg.channels[1].push_back(cur_pts[j].x); msg.channels[2].push_back(cur_pts[j].y);
                msg.channels[3].push_back(pts_velocity[j].x); msg.channels[4].push_back(pts_velocity[j].y);
            }
        }
        msg.channels[5] = get_depth ? get_depth(cur_img_time, msg.points) : std::vector<float>(msg.points.size(), -1.f);   // :215-222
        if (feature_points) *feature_points = std::move(msg);
        if (!init_pub) { init_pub = true; return FIRST_PUBLISH_SUPPRESSED; }              // :225-231 no optical speed on the first image
        return PUBLISHED;
    }
    FeatureTracker& trackerData;
    int pub_count = 1;                          // feature_tracker_node.cpp:19
private:
    int FREQ;
    bool first_image_flag = true, init_pub = false;
    double first_image_time = 0.0, last_image_time = 0.0;
};

}  // namespace lvi_host
