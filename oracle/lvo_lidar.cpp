// TEST INFRASTRUCTURE — CPU oracle of the lidar_odometry hot path (SURVEY §8 a-0 … a-10).
// Exports the same C-ABI as the product library (include/lvi_hotpath.h) so that parity
// tests drive both through one binding.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product never does.
//
// PARITY UNPINNED.  The reference (valentinomario/LiDAR-Visual-Inertial-SLAM) has no tests,
// fixtures or golden vectors, and it cannot be compiled here (needs rclcpp, PCL, OpenCV,
// Eigen, tf2, GTSAM, livox_ros_driver2 — none installed, no network).  This file restates
//   lidar_odometry/src/imageProjection.cpp:239-260, 570-647
//   lidar_odometry/src/featureExtraction.cpp:87-245
//   lidar_odometry/src/mapOptimization.cpp:339-385, 404-407, 958-965, 987-1375
// line by line; the PCL / FLANN / OpenCV / Eigen / tf2 arithmetic those lines call is restated
// from the published algorithms in lvo_voxel.h, lvo_kdtree.h and lvo_math.h.
// Threading mirrors the reference: OpenMP only on the loops it parallelises
// (mapOptimization.cpp:356,375,1010,1102), num_threads(numberOfCores).
#include <omp.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <array>
#include <vector>

#include "../include/lvi_hotpath.h"
#include "lvo_kdtree.h"
#include "lvo_math.h"
#include "lvo_voxel.h"

namespace {

thread_local std::string g_err;
int32_t fail(int32_t code, const char* msg) { g_err = msg; return code; }
}  // namespace
void lvo_set_error(const char* msg) { g_err = msg; }
namespace {

struct smoothness_t { float value; size_t ind; };                       // featureExtraction.cpp:4-7
struct by_value { bool operator()(smoothness_t const& l, smoothness_t const& r) { return l.value < r.value; } };  // :9-13

}  // namespace

struct lvi_lidar {
    lvi_lidar_params P;
    // ---- a-0
    std::vector<lvi_livox_pt> raw;                   // after moveFromCustomMsg (last point dropped)
    bool have_raw = false;
    std::vector<lvi_pt> extracted;                   // extractedCloud
    std::vector<float> range;                        // cloudInfo.point_range
    std::vector<int32_t> col;                        // cloudInfo.point_col_ind
    std::vector<int32_t> startR, endR;
    bool have_org = false;
    // ---- f-4 (cornerCloudKeyFrames, surfCloudKeyFrames, cloudKeyPoses6D: mapOptimization.cpp:76-81)
    std::vector<std::vector<lvi_pt>> cornerCloudKeyFrames, surfCloudKeyFrames;
    std::vector<std::array<float, 6>> keyPoses;
    size_t kf_points = 0;
    // ---- f-1 (imuDeskewInfo's table, imageProjection.cpp:77-80)
    bool imu_available = false;
    int imuPointerCur = 0;
    double timeScanCur = 0.0;
    std::vector<double> imuTime, imuRotX, imuRotY, imuRotZ;
    // ---- a-1..a-3 (arrays persist across scans: featureExtraction.cpp:59,67-69)
    std::vector<smoothness_t> cloudSmoothness;
    std::vector<float> cloudCurvature;
    std::vector<int> cloudNeighborPicked, cloudLabel, pickedOccl;
    std::vector<lvi_pt> corner, surf;
    std::vector<int32_t> corner_index;
    bool have_feat = false;
    // ---- scan DS
    std::vector<lvi_pt> cornerDS, surfDS;
    bool have_ds = false;
    // ---- map
    std::vector<lvi_pt> mapCornerRaw, mapSurfRaw, mapCornerDS, mapSurfDS;
    lvo::KdTree3f kdCorner, kdSurf;
    bool have_map_raw = false, have_map = false;
    // ---- batched entry points: the oracle has one pipeline, a batch is a loop over it
    std::vector<std::vector<lvi_livox_pt>> batch_scans;   // as handed over (point_num entries each)
    std::vector<lvi_pose_record> batch_records;
    // ---- icp
    float T[6] = {0, 0, 0, 0, 0, 0};                 // transformTobeMapped
    bool isDegenerate = false;
    std::vector<lvi_pt> oriCornerVec, coeffCornerVec, oriSurfVec, coeffSurfVec;
    std::vector<uint8_t> flagCorner, flagSurf;       // std::vector<bool> in the reference (Appendix B.9)
    std::vector<lvi_pt> laserCloudOri, coeffSel;
    // ---- debug
    lvo::VoxelDebug vdbg;
    std::vector<float> jtj_trace, pose_trace;
    lvi_pose_record last_record{};
};

namespace {

inline float pointDistance(const lvi_pt& p) { return std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z); }   // utility.h:403-406

// ---------------------------------------------------------------------------
// f-1  findRotation (imageProjection.cpp:495-520) and deskewPoint (:538-568).  findPosition (:522-536) returns
// zeros (its body is commented out in the reference), so every transform here has a zero translation.
// Eigen pieces restated (Eigen 3.4, not in the reference tree, PARITY UNPINNED): Affine3f::inverse() =
// 3x3 cofactor inverse of the linear part (Inverse.h compute_inverse_size3), Affine3f * Affine3f =
// linear * linear with the three products of an element summed left to right.
// ---------------------------------------------------------------------------
void findRotation(const lvi_lidar* h, double pointTime, float* rotXCur, float* rotYCur, float* rotZCur)
{
    *rotXCur = 0; *rotYCur = 0; *rotZCur = 0;
    int imuPointerFront = 0;
    while (imuPointerFront < h->imuPointerCur) {
        if (pointTime < h->imuTime[imuPointerFront]) break;
        ++imuPointerFront;
    }
    if (pointTime > h->imuTime[imuPointerFront] || imuPointerFront == 0) {
        *rotXCur = (float)h->imuRotX[imuPointerFront];
        *rotYCur = (float)h->imuRotY[imuPointerFront];
        *rotZCur = (float)h->imuRotZ[imuPointerFront];
    } else {
        int imuPointerBack = imuPointerFront - 1;
        double ratioFront = (pointTime - h->imuTime[imuPointerBack]) / (h->imuTime[imuPointerFront] - h->imuTime[imuPointerBack]);
        double ratioBack = (h->imuTime[imuPointerFront] - pointTime) / (h->imuTime[imuPointerFront] - h->imuTime[imuPointerBack]);
        *rotXCur = (float)(h->imuRotX[imuPointerFront] * ratioFront + h->imuRotX[imuPointerBack] * ratioBack);
        *rotYCur = (float)(h->imuRotY[imuPointerFront] * ratioFront + h->imuRotY[imuPointerBack] * ratioBack);
        *rotZCur = (float)(h->imuRotZ[imuPointerFront] * ratioFront + h->imuRotZ[imuPointerBack] * ratioBack);
    }
}

struct Mat3f { float m[3][3]; };

inline float cofactor3(const Mat3f& a, int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
}
inline Mat3f inverse3(const Mat3f& a)
{
    const float c0 = cofactor3(a, 0, 0), c1 = cofactor3(a, 1, 0), c2 = cofactor3(a, 2, 0);
    const float det = (c0 * a.m[0][0] + c1 * a.m[1][0]) + c2 * a.m[2][0];
    const float invdet = 1.0f / det;
    Mat3f r;
    r.m[0][0] = c0 * invdet; r.m[0][1] = c1 * invdet; r.m[0][2] = c2 * invdet;
    r.m[1][0] = cofactor3(a, 0, 1) * invdet; r.m[1][1] = cofactor3(a, 1, 1) * invdet; r.m[1][2] = cofactor3(a, 2, 1) * invdet;
    r.m[2][0] = cofactor3(a, 0, 2) * invdet; r.m[2][1] = cofactor3(a, 1, 2) * invdet; r.m[2][2] = cofactor3(a, 2, 2) * invdet;
    return r;
}
inline Mat3f mul3(const Mat3f& a, const Mat3f& b)
{
    Mat3f r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
    return r;
}
inline Mat3f rotationOf(float rotX, float rotY, float rotZ)
{
    const lvo::Affine3f t = lvo::getTransformation(0.f, 0.f, 0.f, rotX, rotY, rotZ);       // posCur = 0 (:522-536)
    Mat3f r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = t.m[i][j];
    return r;
}

// ---------------------------------------------------------------------------
// a-0  imageProjection.cpp:570-647 (sensor == LIVOX)
// ---------------------------------------------------------------------------
void organize(lvi_lidar* h)
{
    bool firstPointFlag = true;                                             // resetParameters
    Mat3f transStartInverse{};
    const lvi_lidar_params& P = h->P;
    const int N_SCAN = P.N_SCAN, H = P.Horizon_SCAN;
    std::vector<float> rangeMat((size_t)N_SCAN * H, FLT_MAX);
    std::vector<lvi_pt> fullCloud((size_t)N_SCAN * H);
    std::vector<int> columnIdnCountVec(N_SCAN, 0);

    const int cloudSize = (int)h->raw.size();
    for (int i = 0; i < cloudSize; ++i) {                                   // projectPointCloud :574
        lvi_pt thisPoint;
        thisPoint.x = h->raw[i].x; thisPoint.y = h->raw[i].y; thisPoint.z = h->raw[i].z;
        thisPoint.intensity = (float)h->raw[i].reflectivity;               // :254
        float range = pointDistance(thisPoint);
        if (range < P.lidarMinRange || range > P.lidarMaxRange) continue;   // :583
        int rowIdn = h->raw[i].line;                                        // :586 (ring = line :257)
        if (rowIdn < 0 || rowIdn >= N_SCAN) continue;
        if (rowIdn % P.downsampleRate != 0) continue;
        int columnIdn = columnIdnCountVec[rowIdn];                          // :604-605
        columnIdnCountVec[rowIdn] += 1;
        if (columnIdn < 0 || columnIdn >= H) continue;                      // :609
        if (rangeMat[(size_t)rowIdn * H + columnIdn] != FLT_MAX) continue;  // :612
        if (h->imu_available) {                                             // deskewPoint :538-568 (deskewFlag stays 0, :109)
            const float relTimeF = (float)(h->raw[i].offset_time * 1e-9);   // PointXYZIRT::time is a float (:255)
            const double pointTime = h->timeScanCur + (double)relTimeF;
            float rotXCur, rotYCur, rotZCur;
            findRotation(h, pointTime, &rotXCur, &rotYCur, &rotZCur);
            const Mat3f transFinal = rotationOf(rotXCur, rotYCur, rotZCur);
            if (firstPointFlag) { transStartInverse = inverse3(transFinal); firstPointFlag = false; }
            const Mat3f transBt = mul3(transStartInverse, transFinal);
            const float tx = 0.f;                                           // lhs.linear * 0 + (-(inv * 0)) = +0
            lvi_pt np;
            np.x = transBt.m[0][0] * thisPoint.x + transBt.m[0][1] * thisPoint.y + transBt.m[0][2] * thisPoint.z + tx;
            np.y = transBt.m[1][0] * thisPoint.x + transBt.m[1][1] * thisPoint.y + transBt.m[1][2] * thisPoint.z + tx;
            np.z = transBt.m[2][0] * thisPoint.x + transBt.m[2][1] * thisPoint.y + transBt.m[2][2] * thisPoint.z + tx;
            np.intensity = thisPoint.intensity;
            thisPoint = np;
        }
        rangeMat[(size_t)rowIdn * H + columnIdn] = range;
        fullCloud[(size_t)columnIdn + (size_t)rowIdn * H] = thisPoint;      // :619-620
    }
    h->extracted.clear(); h->range.clear(); h->col.clear();
    h->startR.assign(N_SCAN, 0); h->endR.assign(N_SCAN, 0);
    int count = 0;
    for (int i = 0; i < N_SCAN; ++i) {                                      // cloudExtraction :628
        h->startR[i] = count - 1 + 5;
        for (int j = 0; j < H; ++j) {
            if (rangeMat[(size_t)i * H + j] != FLT_MAX) {
                h->col.push_back(j);
                h->range.push_back(rangeMat[(size_t)i * H + j]);
                h->extracted.push_back(fullCloud[(size_t)j + (size_t)i * H]);
                ++count;
            }
        }
        h->endR[i] = count - 1 - 5;
    }
    h->have_org = true;
}

// ---------------------------------------------------------------------------
// a-1  featureExtraction.cpp:87-111
// ---------------------------------------------------------------------------
void calculateSmoothness(lvi_lidar* h)
{
    const std::vector<float>& point_range = h->range;
    int cloudSize = (int)h->extracted.size();
    for (int i = 5; i < cloudSize - 5; i++) {
        float diffRange = point_range[i - 2] + point_range[i - 1] - point_range[i] * 4
                        + point_range[i + 1] + point_range[i + 2];
        h->cloudCurvature[i] = diffRange * diffRange;
        h->cloudNeighborPicked[i] = 0;
        h->cloudLabel[i] = 0;
        h->cloudSmoothness[i].value = h->cloudCurvature[i];
        h->cloudSmoothness[i].ind = i;
    }
}

// ---------------------------------------------------------------------------
// a-2  featureExtraction.cpp:113-148
// ---------------------------------------------------------------------------
void markOccludedPoints(lvi_lidar* h)
{
    const std::vector<float>& point_range = h->range;
    const std::vector<int32_t>& point_col_ind = h->col;
    int* cloudNeighborPicked = h->cloudNeighborPicked.data();
    int cloudSize = (int)h->extracted.size();
    for (int i = 5; i < cloudSize - 6; ++i) {
        float depth1 = point_range[i];
        float depth2 = point_range[i + 1];
        int columnDiff = std::abs(int(point_col_ind[i + 1] - point_col_ind[i]));
        if (columnDiff < 10) {
            if (depth1 - depth2 > 0.3) {                 // float - float promoted to double against 0.3
                cloudNeighborPicked[i - 1] = 1;
                cloudNeighborPicked[i] = 1;
            } else if (depth2 - depth1 > 0.3) {
                cloudNeighborPicked[i + 1] = 1;
                cloudNeighborPicked[i + 2] = 1;
            }
        }
        float diff1 = std::abs(float(point_range[i - 1] - point_range[i]));
        float diff2 = std::abs(float(point_range[i + 1] - point_range[i]));
        if (diff1 > 0.1 * point_range[i] && diff2 > 0.1 * point_range[i])   // 0.1 * float is a double product
            cloudNeighborPicked[i] = 1;
    }
}

// ---------------------------------------------------------------------------
// a-3  featureExtraction.cpp:150-245 (+ per-ring VoxelGrid a-4)
// ---------------------------------------------------------------------------
void extractFeatures(lvi_lidar* h)
{
    const lvi_lidar_params& P = h->P;
    const std::vector<int32_t>& point_col_ind = h->col;
    int* cloudNeighborPicked = h->cloudNeighborPicked.data();
    int* cloudLabel = h->cloudLabel.data();
    float* cloudCurvature = h->cloudCurvature.data();
    std::vector<smoothness_t>& cloudSmoothness = h->cloudSmoothness;
    const std::vector<lvi_pt>& pts = h->extracted;

    h->corner.clear(); h->surf.clear(); h->corner_index.clear();
    std::vector<lvi_pt> surfaceCloudScan, surfaceCloudScanDS;

    for (int i = 0; i < P.N_SCAN; i++) {
        surfaceCloudScan.clear();
        for (int j = 0; j < 6; j++) {
            int sp = (h->startR[i] * (6 - j) + h->endR[i] * j) / 6;
            int ep = (h->startR[i] * (5 - j) + h->endR[i] * (j + 1)) / 6 - 1;
            if (sp >= ep) continue;

            std::sort(cloudSmoothness.begin() + sp, cloudSmoothness.begin() + ep, by_value());   // half-open: element ep unsorted

            int largestPickedNum = 0;
            for (int k = ep; k >= sp; k--) {
                int ind = (int)cloudSmoothness[k].ind;
                if (cloudNeighborPicked[ind] == 0 && cloudCurvature[ind] > P.edgeThreshold) {
                    largestPickedNum++;
                    if (largestPickedNum <= 40) {
                        cloudLabel[ind] = 1;
                        h->corner.push_back(pts[ind]);
                        h->corner_index.push_back(ind);
                    } else {
                        break;
                    }
                    cloudNeighborPicked[ind] = 1;
                    for (int l = 1; l <= 5; l++) {
                        int columnDiff = std::abs(int(point_col_ind[ind + l] - point_col_ind[ind + l - 1]));
                        if (columnDiff > 10) break;
                        cloudNeighborPicked[ind + l] = 1;
                    }
                    for (int l = -1; l >= -5; l--) {
                        int columnDiff = std::abs(int(point_col_ind[ind + l] - point_col_ind[ind + l + 1]));
                        if (columnDiff > 10) break;
                        cloudNeighborPicked[ind + l] = 1;
                    }
                }
            }

            for (int k = sp; k <= ep; k++) {
                int ind = (int)cloudSmoothness[k].ind;
                if (cloudNeighborPicked[ind] == 0 && cloudCurvature[ind] < P.surfThreshold) {
                    cloudLabel[ind] = -1;
                    cloudNeighborPicked[ind] = 1;
                    for (int l = 1; l <= 5; l++) {
                        // ind == 0 happens only for the never-rewritten slot cloudSmoothness[4] = {0,0}
                        // (SURVEY Appendix B.4); the reference then reads point_col_ind[-1] (UB).
                        // Deviation: treat an out-of-range neighbour as a column break.
                        if (ind + l >= (int)point_col_ind.size()) break;
                        int columnDiff = std::abs(int(point_col_ind[ind + l] - point_col_ind[ind + l - 1]));
                        if (columnDiff > 10) break;
                        cloudNeighborPicked[ind + l] = 1;
                    }
                    for (int l = -1; l >= -5; l--) {
                        if (ind + l < 0) break;
                        int columnDiff = std::abs(int(point_col_ind[ind + l] - point_col_ind[ind + l + 1]));
                        if (columnDiff > 10) break;
                        cloudNeighborPicked[ind + l] = 1;
                    }
                }
            }

            for (int k = sp; k <= ep; k++) {
                if (cloudLabel[k] <= 0) surfaceCloudScan.push_back(pts[k]);
            }
        }
        surfaceCloudScanDS.clear();
        lvo::voxel_grid_filter(surfaceCloudScan.data(), (int)surfaceCloudScan.size(), P.odometrySurfLeafSize, surfaceCloudScanDS);   // :239-241
        h->surf.insert(h->surf.end(), surfaceCloudScanDS.begin(), surfaceCloudScanDS.end());   // :243
    }
}

void extract(lvi_lidar* h)
{
    // The reference keeps these arrays for the life of the node and never clears them;
    // a fresh node sees zero-filled heap pages (documented assumption).
    const size_t full = (size_t)h->P.N_SCAN * h->P.Horizon_SCAN;
    if (h->cloudSmoothness.size() != full) {
        h->cloudSmoothness.assign(full, smoothness_t{0.f, 0});
        h->cloudCurvature.assign(full, 0.f);
        h->cloudNeighborPicked.assign(full, 0);
        h->cloudLabel.assign(full, 0);
    }
    calculateSmoothness(h);
    markOccludedPoints(h);
    h->pickedOccl.assign(h->cloudNeighborPicked.begin(), h->cloudNeighborPicked.begin() + h->extracted.size());
    extractFeatures(h);
    h->have_feat = true;
}

// ---------------------------------------------------------------------------
// a-5  mapOptimization.cpp:347-385
// ---------------------------------------------------------------------------
void transformPointCloud(const lvi_pt* in, int n, const lvo::Affine3f& t, lvi_pt* out, int threads)
{
#pragma omp parallel for num_threads(threads)
    for (int i = 0; i < n; ++i) {
        const lvi_pt& p = in[i];
        lvi_pt o;
        o.x = t.m[0][0] * p.x + t.m[0][1] * p.y + t.m[0][2] * p.z + t.m[0][3];
        o.y = t.m[1][0] * p.x + t.m[1][1] * p.y + t.m[1][2] * p.z + t.m[1][3];
        o.z = t.m[2][0] * p.x + t.m[2][1] * p.y + t.m[2][2] * p.z + t.m[2][3];
        o.intensity = p.intensity;
        out[i] = o;
    }
}

inline void pointAssociateToMap(const lvo::Affine3f& t, const lvi_pt* pi, lvi_pt* po)   // :339-345
{
    po->x = t.m[0][0] * pi->x + t.m[0][1] * pi->y + t.m[0][2] * pi->z + t.m[0][3];
    po->y = t.m[1][0] * pi->x + t.m[1][1] * pi->y + t.m[1][2] * pi->z + t.m[1][3];
    po->z = t.m[2][0] * pi->x + t.m[2][1] * pi->y + t.m[2][2] * pi->z + t.m[2][3];
    po->intensity = pi->intensity;
}

// ---------------------------------------------------------------------------
// a-7  cornerOptimization mapOptimization.cpp:1006-1096, one point
// ---------------------------------------------------------------------------
inline bool cornerResidual(const lvi_lidar* h, const lvo::Affine3f& T, const lvi_pt& pointOri, lvi_pt& coeff)
{
    lvi_pt pointSel;
    pointAssociateToMap(T, &pointOri, &pointSel);
    int pointSearchInd[5]; float pointSearchSqDis[5];
    const float q[3] = {pointSel.x, pointSel.y, pointSel.z};
    int found = h->kdCorner.knn(q, 5, pointSearchInd, pointSearchSqDis);
    if (found < 5) return false;          // reference would index out of range; unreachable past the feature gates
    const std::vector<lvi_pt>& map = h->mapCornerDS;
    if (pointSearchSqDis[4] < 1.0) {
        float cx = 0, cy = 0, cz = 0;
        for (int j = 0; j < 5; j++) { cx += map[pointSearchInd[j]].x; cy += map[pointSearchInd[j]].y; cz += map[pointSearchInd[j]].z; }
        cx /= 5; cy /= 5; cz /= 5;
        float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
        for (int j = 0; j < 5; j++) {
            float ax = map[pointSearchInd[j]].x - cx;
            float ay = map[pointSearchInd[j]].y - cy;
            float az = map[pointSearchInd[j]].z - cz;
            a11 += ax * ax; a12 += ax * ay; a13 += ax * az;
            a22 += ay * ay; a23 += ay * az;
            a33 += az * az;
        }
        a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
        float matA1[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33};
        float matD1[3], matV1[9];
        lvo::jacobi_f32(matA1, 3, matD1, matV1);                      // cv::eigen :1050
        if (matD1[0] > 3 * matD1[1]) {
            float x0 = pointSel.x, y0 = pointSel.y, z0 = pointSel.z;
            float x1 = cx + 0.1 * matV1[0];
            float y1 = cy + 0.1 * matV1[1];
            float z1 = cz + 0.1 * matV1[2];
            float x2 = cx - 0.1 * matV1[0];
            float y2 = cy - 0.1 * matV1[1];
            float z2 = cz - 0.1 * matV1[2];
            float a012 = std::sqrt(((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1))
                                 + ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))
                                 + ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)));
            float l12 = std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
            float la = ((y1 - y2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1))
                      + (z1 - z2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))) / a012 / l12;
            float lb = -((x1 - x2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1))
                       - (z1 - z2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
            float lc = -((x1 - x2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))
                       + (y1 - y2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
            float ld2 = a012 / l12;
            float s = 1 - 0.9 * std::fabs(ld2);
            coeff.x = s * la; coeff.y = s * lb; coeff.z = s * lc; coeff.intensity = s * ld2;
            if (s > 0.1) return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------
// a-8  surfOptimization mapOptimization.cpp:1098-1167, one point
// ---------------------------------------------------------------------------
inline bool surfResidual(const lvi_lidar* h, const lvo::Affine3f& T, const lvi_pt& pointOri, lvi_pt& coeff)
{
    lvi_pt pointSel;
    pointAssociateToMap(T, &pointOri, &pointSel);
    int pointSearchInd[5]; float pointSearchSqDis[5];
    const float q[3] = {pointSel.x, pointSel.y, pointSel.z};
    int found = h->kdSurf.knn(q, 5, pointSearchInd, pointSearchSqDis);
    if (found < 5) return false;
    const std::vector<lvi_pt>& map = h->mapSurfDS;
    if (pointSearchSqDis[4] < 1.0) {
        float matA0[5][3], matB0[5], matX0[3];
        for (int j = 0; j < 5; j++) {
            matA0[j][0] = map[pointSearchInd[j]].x;
            matA0[j][1] = map[pointSearchInd[j]].y;
            matA0[j][2] = map[pointSearchInd[j]].z;
            matB0[j] = -1.f;
        }
        lvo::colpiv_qr_solve_5x3(matA0, matB0, matX0);               // :1128
        float pa = matX0[0], pb = matX0[1], pc = matX0[2], pd = 1;
        float ps = std::sqrt(pa * pa + pb * pb + pc * pc);
        pa /= ps; pb /= ps; pc /= ps; pd /= ps;
        bool planeValid = true;
        for (int j = 0; j < 5; j++) {
            if (std::fabs(pa * map[pointSearchInd[j]].x + pb * map[pointSearchInd[j]].y + pc * map[pointSearchInd[j]].z + pd) > 0.2) {
                planeValid = false;
                break;
            }
        }
        if (planeValid) {
            float pd2 = pa * pointSel.x + pb * pointSel.y + pc * pointSel.z + pd;
            float s = 1 - 0.9 * std::fabs(pd2) / std::sqrt(std::sqrt(pointOri.x * pointOri.x + pointOri.y * pointOri.y + pointOri.z * pointOri.z));
            coeff.x = s * pa; coeff.y = s * pb; coeff.z = s * pc; coeff.intensity = s * pd2;
            if (s > 0.1) return true;
        }
    }
    return false;
}

void cornerOptimization(lvi_lidar* h)
{
    const lvo::Affine3f T = lvo::trans2Affine3f(h->T);               // updatePointAssociateToMap :1001-1004
    const int n = (int)h->cornerDS.size();
#pragma omp parallel for num_threads(h->P.numberOfCores)
    for (int i = 0; i < n; i++) {
        lvi_pt coeff;
        if (cornerResidual(h, T, h->cornerDS[i], coeff)) {
            h->oriCornerVec[i] = h->cornerDS[i];
            h->coeffCornerVec[i] = coeff;
            h->flagCorner[i] = 1;
        }
    }
}

void surfOptimization(lvi_lidar* h)
{
    const lvo::Affine3f T = lvo::trans2Affine3f(h->T);
    const int n = (int)h->surfDS.size();
#pragma omp parallel for num_threads(h->P.numberOfCores)
    for (int i = 0; i < n; i++) {
        lvi_pt coeff;
        if (surfResidual(h, T, h->surfDS[i], coeff)) {
            h->oriSurfVec[i] = h->surfDS[i];
            h->coeffSurfVec[i] = coeff;
            h->flagSurf[i] = 1;
        }
    }
}

void combineOptimizationCoeffs(lvi_lidar* h)                          // :1169-1188
{
    for (size_t i = 0; i < h->cornerDS.size(); ++i)
        if (h->flagCorner[i]) { h->laserCloudOri.push_back(h->oriCornerVec[i]); h->coeffSel.push_back(h->coeffCornerVec[i]); }
    for (size_t i = 0; i < h->surfDS.size(); ++i)
        if (h->flagSurf[i]) { h->laserCloudOri.push_back(h->oriSurfVec[i]); h->coeffSel.push_back(h->coeffSurfVec[i]); }
    std::fill(h->flagCorner.begin(), h->flagCorner.end(), 0);
    std::fill(h->flagSurf.begin(), h->flagSurf.end(), 0);
}

// ---------------------------------------------------------------------------
// a-9  LMOptimization mapOptimization.cpp:1190-1313
// ---------------------------------------------------------------------------
bool LMOptimization(lvi_lidar* h, int iterCount)
{
    float* transformTobeMapped = h->T;
    float srx = std::sin(transformTobeMapped[1]);
    float crx = std::cos(transformTobeMapped[1]);
    float sry = std::sin(transformTobeMapped[2]);
    float cry = std::cos(transformTobeMapped[2]);
    float srz = std::sin(transformTobeMapped[0]);
    float crz = std::cos(transformTobeMapped[0]);

    int laserCloudSelNum = (int)h->laserCloudOri.size();
    if (laserCloudSelNum < 50) return false;

    std::vector<float> matA((size_t)laserCloudSelNum * 6), matB(laserCloudSelNum);
    for (int i = 0; i < laserCloudSelNum; i++) {
        lvi_pt pointOri, coeff;
        pointOri.x = h->laserCloudOri[i].y;
        pointOri.y = h->laserCloudOri[i].z;
        pointOri.z = h->laserCloudOri[i].x;
        coeff.x = h->coeffSel[i].y;
        coeff.y = h->coeffSel[i].z;
        coeff.z = h->coeffSel[i].x;
        coeff.intensity = h->coeffSel[i].intensity;
        float arx = (crx * sry * srz * pointOri.x + crx * crz * sry * pointOri.y - srx * sry * pointOri.z) * coeff.x
                  + (-srx * srz * pointOri.x - crz * srx * pointOri.y - crx * pointOri.z) * coeff.y
                  + (crx * cry * srz * pointOri.x + crx * cry * crz * pointOri.y - cry * srx * pointOri.z) * coeff.z;
        float ary = ((cry * srx * srz - crz * sry) * pointOri.x
                  + (sry * srz + cry * crz * srx) * pointOri.y + crx * cry * pointOri.z) * coeff.x
                  + ((-cry * crz - srx * sry * srz) * pointOri.x
                  + (cry * srz - crz * srx * sry) * pointOri.y - crx * sry * pointOri.z) * coeff.z;
        float arz = ((crz * srx * sry - cry * srz) * pointOri.x + (-cry * crz - srx * sry * srz) * pointOri.y) * coeff.x
                  + (crx * crz * pointOri.x - crx * srz * pointOri.y) * coeff.y
                  + ((sry * srz + cry * crz * srx) * pointOri.x + (crz * sry - cry * srx * srz) * pointOri.y) * coeff.z;
        matA[(size_t)i * 6 + 0] = arz;
        matA[(size_t)i * 6 + 1] = arx;
        matA[(size_t)i * 6 + 2] = ary;
        matA[(size_t)i * 6 + 3] = coeff.z;
        matA[(size_t)i * 6 + 4] = coeff.x;
        matA[(size_t)i * 6 + 5] = coeff.y;
        matB[i] = -coeff.intensity;
    }
    // matAtA = matAt * matA; matAtB = matAt * matB  (cv::gemm, f32 storage, double accumulation)
    float matAtA[36], matAtB[6], matX[6];
    for (int r = 0; r < 6; r++) {
        for (int c = 0; c < 6; c++) {
            double s = 0;
            for (int i = 0; i < laserCloudSelNum; i++) s += (double)matA[(size_t)i * 6 + r] * (double)matA[(size_t)i * 6 + c];
            matAtA[r * 6 + c] = (float)s;
        }
        double s = 0;
        for (int i = 0; i < laserCloudSelNum; i++) s += (double)matA[(size_t)i * 6 + r] * (double)matB[i];
        matAtB[r] = (float)s;
    }
    {   // debug trace: 21 upper-triangular + 6
        int k = 0; float rec[27];
        for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) rec[k++] = matAtA[r * 6 + c];
        for (int r = 0; r < 6; r++) rec[k++] = matAtB[r];
        h->jtj_trace.insert(h->jtj_trace.end(), rec, rec + 27);
    }
    {
        float Acopy[36]; std::memcpy(Acopy, matAtA, sizeof(Acopy));
        std::memcpy(matX, matAtB, sizeof(matX));
        if (!lvo::solve_qr_f32(Acopy, 6, matX)) std::memset(matX, 0, sizeof(matX));   // cv::solve returns false, dst zeroed
    }
    float matP[36];                                            // local, shadows the member (Appendix B.10)
    std::memset(matP, 0, sizeof(matP));
    if (iterCount == 0) {
        float matE[6], matV[36], matV2[36], Acopy[36];
        std::memcpy(Acopy, matAtA, sizeof(Acopy));
        lvo::jacobi_f32(Acopy, 6, matE, matV);                 // cv::eigen :1268
        std::memcpy(matV2, matV, sizeof(matV2));
        h->isDegenerate = false;
        float eignThre[6] = {100, 100, 100, 100, 100, 100};
        for (int i = 5; i >= 0; i--) {
            if (matE[i] < eignThre[i]) {
                for (int j = 0; j < 6; j++) matV2[i * 6 + j] = 0;
                h->isDegenerate = true;
            } else {
                break;
            }
        }
        float Vinv[36];
        lvo::inv_lu_f32(matV, 6, Vinv);
        lvo::gemm_f32_dacc(Vinv, matV2, matP, 6, 6, 6);        // matP = matV.inv() * matV2
    }
    if (h->isDegenerate) {
        float matX2[6]; std::memcpy(matX2, matX, sizeof(matX2));
        lvo::gemm_f32_dacc(matP, matX2, matX, 6, 6, 1);        // iterations >= 1: matP is all zeros
    }
    transformTobeMapped[0] += matX[0];
    transformTobeMapped[1] += matX[1];
    transformTobeMapped[2] += matX[2];
    transformTobeMapped[3] += matX[3];
    transformTobeMapped[4] += matX[4];
    transformTobeMapped[5] += matX[5];

    // pcl::rad2deg(float) = alpha * 57.29578f; pow(float,int) promotes to double
    float deltaR = std::sqrt(std::pow(matX[0] * 57.29578f, 2) + std::pow(matX[1] * 57.29578f, 2) + std::pow(matX[2] * 57.29578f, 2));
    float deltaT = std::sqrt(std::pow(matX[3] * 100, 2) + std::pow(matX[4] * 100, 2) + std::pow(matX[5] * 100, 2));
    if (deltaR < 0.05 && deltaT < 0.05) return true;
    return false;
}

inline float constraintTransformation(float value, float limit)      // :1377-1385
{
    if (value < -limit) value = -limit;
    if (value > limit) value = limit;
    return value;
}

void transformUpdate(lvi_lidar* h, const lvi_imu_hint* imu)          // :1345-1375
{
    float* T = h->T;
    if (imu && imu->imu_available) {
        if (std::abs(imu->imu_pitch_init) < 1.4) {
            double imuWeight = h->P.imuRPYWeight;
            double rollMid, pitchMid, yawMid;
            lvo::Quat tq = lvo::quat_setRPY(T[0], 0, 0);
            lvo::Quat iq = lvo::quat_setRPY(imu->imu_roll_init, 0, 0);
            lvo::quat_getRPY(lvo::quat_slerp(tq, iq, imuWeight), rollMid, pitchMid, yawMid);
            T[0] = (float)rollMid;
            tq = lvo::quat_setRPY(0, T[1], 0);
            iq = lvo::quat_setRPY(0, imu->imu_pitch_init, 0);
            lvo::quat_getRPY(lvo::quat_slerp(tq, iq, imuWeight), rollMid, pitchMid, yawMid);
            T[1] = (float)pitchMid;
        }
    }
    T[0] = constraintTransformation(T[0], h->P.rotation_tollerance);
    T[1] = constraintTransformation(T[1], h->P.rotation_tollerance);
    T[5] = constraintTransformation(T[5], h->P.z_tollerance);
}

void downsampleCurrentScan(lvi_lidar* h)                               // :987-999
{
    lvo::voxel_grid_filter(h->corner.data(), (int)h->corner.size(), h->P.mappingCornerLeafSize, h->cornerDS);
    lvo::voxel_grid_filter(h->surf.data(), (int)h->surf.size(), h->P.mappingSurfLeafSize, h->surfDS);
    h->have_ds = true;
}

void mapBuild(lvi_lidar* h)                                            // :958-965 + :1322-1323
{
    lvo::voxel_grid_filter(h->mapCornerRaw.data(), (int)h->mapCornerRaw.size(), h->P.mappingCornerLeafSize, h->mapCornerDS);
    lvo::voxel_grid_filter(h->mapSurfRaw.data(), (int)h->mapSurfRaw.size(), h->P.mappingSurfLeafSize, h->mapSurfDS);
    h->kdCorner.build(&h->mapCornerDS.data()->x, 4, (int)h->mapCornerDS.size());
    h->kdSurf.build(&h->mapSurfDS.data()->x, 4, (int)h->mapSurfDS.size());
    h->have_map = true;
}

int32_t scan2MapOptimization(lvi_lidar* h, const lvi_imu_hint* imu, lvi_icp_result* out)   // :1315-1343
{
    std::memset(out, 0, sizeof(*out));
    h->jtj_trace.clear(); h->pose_trace.clear();
    const int nC = (int)h->cornerDS.size(), nS = (int)h->surfDS.size();
    out->n_corner_ds = nC; out->n_surf_ds = nS;
    if (!h->have_map) { out->status = LVI_NO_MAP; std::memcpy(out->pose, h->T, sizeof(float) * 6); return LVI_NO_MAP; }
    if (nC > h->P.edgeFeatureMinValidNum && nS > h->P.surfFeatureMinValidNum) {
        h->oriCornerVec.resize(nC); h->coeffCornerVec.resize(nC); h->flagCorner.assign(nC, 0);
        h->oriSurfVec.resize(nS); h->coeffSurfVec.resize(nS); h->flagSurf.assign(nS, 0);
        const int maxIters = std::min(h->P.icp_max_iters, LVI_ICP_MAX_ITERS);
        bool any_lm = false;
        for (int iterCount = 0; iterCount < maxIters; iterCount++) {
            h->laserCloudOri.clear(); h->coeffSel.clear();
            h->pose_trace.insert(h->pose_trace.end(), h->T, h->T + 6);
            cornerOptimization(h);
            surfOptimization(h);
            combineOptimizationCoeffs(h);
            out->n_sel[iterCount] = (int)h->laserCloudOri.size();
            out->iters = iterCount + 1;
            if (out->n_sel[iterCount] >= 50) any_lm = true;
            if (LMOptimization(h, iterCount) == true) {
                out->converged = 1;
                if (!h->P.icp_disable_break) break;
            }
        }
        h->pose_trace.insert(h->pose_trace.end(), h->T, h->T + 6);
        out->degenerate = h->isDegenerate ? 1 : 0;
        transformUpdate(h, imu);
        out->status = any_lm ? LVI_OK : LVI_TOO_FEW_CORRESPONDENCES;
    } else {
        out->status = LVI_TOO_FEW_FEATURES;
    }
    std::memcpy(out->pose, h->T, sizeof(float) * 6);
    std::memcpy(h->last_record.pose, h->T, sizeof(float) * 6);
    h->last_record.status = out->status; h->last_record.iters = out->iters;
    return out->status;
}

int32_t copy_cloud(const std::vector<lvi_pt>& src, lvi_cloud* dst)
{
    if (!dst) return LVI_OK;
    dst->n = (int32_t)src.size();
    if (dst->capacity < dst->n || (!dst->pts && dst->n > 0)) return fail(LVI_ERR_CAPACITY, "cloud capacity too small");
    if (dst->n) std::memcpy(dst->pts, src.data(), sizeof(lvi_pt) * src.size());
    return LVI_OK;
}

template <class T>
int32_t dbg_copy(const T* src, size_t n, void* dst, int64_t cap, int64_t* n_bytes)
{
    int64_t bytes = (int64_t)(n * sizeof(T));
    if (n_bytes) *n_bytes = bytes;
    if (!dst) return LVI_OK;
    if (cap < bytes) return fail(LVI_ERR_CAPACITY, "debug buffer too small");
    if (bytes) std::memcpy(dst, src, (size_t)bytes);
    return LVI_OK;
}

}  // namespace

// ===========================================================================
extern "C" {

int32_t lvi_abi_version(void) { return LVI_ABI_VERSION; }
const char* lvi_backend(void) { return "cpu-oracle"; }
const char* lvi_last_error(void) { return g_err.c_str(); }

void lvi_lidar_params_default(lvi_lidar_params* p)
{
    std::memset(p, 0, sizeof(*p));
    p->N_SCAN = 4; p->Horizon_SCAN = 6000; p->downsampleRate = 1;
    p->lidarMinRange = 1.0f; p->lidarMaxRange = 100.0f;
    p->edgeThreshold = 1.0f; p->surfThreshold = 0.1f;
    p->edgeFeatureMinValidNum = 10; p->surfFeatureMinValidNum = 100;
    p->odometrySurfLeafSize = 0.4f; p->mappingCornerLeafSize = 0.2f; p->mappingSurfLeafSize = 0.4f;
    p->z_tollerance = 1000.0f; p->rotation_tollerance = 1000.0f; p->imuRPYWeight = 0.01f;
    p->numberOfCores = 8;
    p->icp_max_iters = 20; p->icp_disable_break = 0;
    p->max_raw_points = 131072; p->max_map_points = 1 << 20; p->voxel_mode = 0;
    p->max_keyframes = 1024; p->max_keyframe_points = 1 << 22; p->map_on_main_stream = 0;
    p->sector_handover_wait_us = 0; p->batch_scans = 1; p->map_plan_cache = 0;
}

int32_t lvi_lidar_create(const lvi_lidar_params* p, int32_t /*device*/, lvi_lidar** out)
{
    if (!p || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (p->N_SCAN <= 0 || p->Horizon_SCAN <= 0 || p->downsampleRate <= 0) return fail(LVI_ERR_INVALID_ARG, "bad scan geometry");
    lvi_lidar* h = new lvi_lidar();
    h->P = *p;
    if (h->P.numberOfCores <= 0) h->P.numberOfCores = 1;
    *out = h;
    return LVI_OK;
}
void lvi_lidar_destroy(lvi_lidar* h) { delete h; }
int32_t lvi_lidar_sync(lvi_lidar*) { return LVI_OK; }
int32_t lvi_lidar_mark(lvi_lidar* h, int32_t slot) { return (h && slot >= 0 && slot < LVI_LIDAR_MARKS) ? LVI_OK : fail(LVI_ERR_INVALID_ARG, "bad mark"); }
int32_t lvi_lidar_wait_mark(lvi_lidar* h, int32_t slot) { return (h && slot >= 0 && slot < LVI_LIDAR_MARKS) ? LVI_OK : fail(LVI_ERR_INVALID_ARG, "bad mark"); }

int32_t lvi_scan_upload(lvi_lidar* h, const lvi_livox_pt* pts, int32_t n_raw)
{
    if (!h || (n_raw > 0 && !pts)) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (n_raw > h->P.max_raw_points) return fail(LVI_ERR_CAPACITY, "n_raw exceeds max_raw_points");
    int n = n_raw > 0 ? n_raw - 1 : 0;                    // moveFromCustomMsg: i < point_num-1  (imageProjection.cpp:249)
    h->raw.assign(pts, pts + n);
    h->have_raw = true; h->have_org = h->have_feat = h->have_ds = false;
    return LVI_OK;
}
int32_t lvi_scan_organize(lvi_lidar* h)
{
    if (!h || !h->have_raw) return fail(LVI_ERR_STATE, "no scan uploaded");
    organize(h);
    h->have_feat = h->have_ds = false;
    return LVI_OK;
}
int32_t lvi_scan_extract(lvi_lidar* h)
{
    if (!h || !h->have_org) return fail(LVI_ERR_STATE, "scan not organised");
    extract(h);
    h->have_ds = false;
    return LVI_OK;
}
int32_t lvi_scan_downsample(lvi_lidar* h)
{
    if (!h || !h->have_feat) return fail(LVI_ERR_STATE, "features not extracted");
    downsampleCurrentScan(h);
    return LVI_OK;
}
int32_t lvi_map_upload(lvi_lidar* h, const lvi_pt* c, int32_t nc, const lvi_pt* s, int32_t ns)
{
    if (!h || (nc > 0 && !c) || (ns > 0 && !s) || nc < 0 || ns < 0) return fail(LVI_ERR_INVALID_ARG, "bad map arguments");
    if (nc > h->P.max_map_points || ns > h->P.max_map_points) return fail(LVI_ERR_CAPACITY, "map exceeds max_map_points");
    h->mapCornerRaw.assign(c, c + nc); h->mapSurfRaw.assign(s, s + ns);
    h->have_map_raw = true; h->have_map = false;
    return LVI_OK;
}
int32_t lvi_map_build(lvi_lidar* h)
{
    if (!h || !h->have_map_raw) return fail(LVI_ERR_STATE, "no map uploaded");
    mapBuild(h);
    return LVI_OK;
}
int32_t lvi_scan_match(lvi_lidar* h, const lvi_imu_hint* imu, float pose[6], lvi_icp_result* out)
{
    if (!h || !pose || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->have_ds) return fail(LVI_ERR_STATE, "scan not downsampled");
    std::memcpy(h->T, pose, sizeof(float) * 6);
    int32_t st = scan2MapOptimization(h, imu, out);
    std::memcpy(pose, h->T, sizeof(float) * 6);
    return st;
}

int32_t lvi_get_scan_info(lvi_lidar* h, lvi_scan_info* out)
{
    if (!h || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->have_org) return fail(LVI_ERR_STATE, "scan not organised");
    out->n = (int32_t)h->extracted.size();
    if (out->capacity < out->n) return fail(LVI_ERR_CAPACITY, "scan_info capacity too small");
    for (int i = 0; i < h->P.N_SCAN; i++) { out->start_ring_index[i] = h->startR[i]; out->end_ring_index[i] = h->endR[i]; }
    if (out->n) {
        std::memcpy(out->point_col_ind, h->col.data(), sizeof(int32_t) * out->n);
        std::memcpy(out->point_range, h->range.data(), sizeof(float) * out->n);
        std::memcpy(out->cloud_deskewed, h->extracted.data(), sizeof(lvi_pt) * out->n);
    }
    return LVI_OK;
}
int32_t lvi_get_features(lvi_lidar* h, lvi_cloud* corner, lvi_cloud* surf)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->have_feat) return fail(LVI_ERR_STATE, "features not extracted");
    int32_t st = copy_cloud(h->corner, corner); if (st) return st;
    return copy_cloud(h->surf, surf);
}
int32_t lvi_get_scan_ds(lvi_lidar* h, lvi_cloud* c, lvi_cloud* s)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->have_ds) return fail(LVI_ERR_STATE, "scan not downsampled");
    int32_t st = copy_cloud(h->cornerDS, c); if (st) return st;
    return copy_cloud(h->surfDS, s);
}
int32_t lvi_get_map_ds(lvi_lidar* h, lvi_cloud* c, lvi_cloud* s)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->have_map) return fail(LVI_ERR_STATE, "map not built");
    int32_t st = copy_cloud(h->mapCornerDS, c); if (st) return st;
    return copy_cloud(h->mapSurfDS, s);
}
int32_t lvi_get_counts(lvi_lidar* h, int32_t counts[8])
{
    if (!h || !counts) return fail(LVI_ERR_INVALID_ARG, "null argument");
    counts[0] = h->have_org ? (int32_t)h->extracted.size() : 0;
    counts[1] = h->have_feat ? (int32_t)h->corner.size() : 0;
    counts[2] = h->have_feat ? (int32_t)h->surf.size() : 0;
    counts[3] = h->have_ds ? (int32_t)h->cornerDS.size() : 0;
    counts[4] = h->have_ds ? (int32_t)h->surfDS.size() : 0;
    counts[5] = h->have_map ? (int32_t)h->mapCornerDS.size() : 0;
    counts[6] = h->have_map ? (int32_t)h->mapSurfDS.size() : 0;
    counts[7] = 0;
    return LVI_OK;
}

int32_t lvi_get_pose_record(lvi_lidar* h, lvi_pose_record* out)
{
    if (!h || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    *out = h->last_record;
    return LVI_OK;
}

// ---- f-4 ---------------------------------------------------------------------
int32_t lvi_keyframe_add(lvi_lidar* h, const lvi_pt* corner, int32_t nc, const lvi_pt* surf, int32_t ns, const float pose[6], int32_t* index_out)
{
    if (!h || nc < 0 || ns < 0 || (nc > 0 && !corner) || (ns > 0 && !surf) || !pose) return fail(LVI_ERR_INVALID_ARG, "bad keyframe arguments");
    if ((int)h->keyPoses.size() >= h->P.max_keyframes || h->kf_points + (size_t)nc + (size_t)ns > (size_t)h->P.max_keyframe_points)
        return fail(LVI_ERR_CAPACITY, "keyframe store full");
    h->cornerCloudKeyFrames.emplace_back(corner, corner + nc);               // :1598
    h->surfCloudKeyFrames.emplace_back(surf, surf + ns);                     // :1599
    h->keyPoses.push_back({pose[0], pose[1], pose[2], pose[3], pose[4], pose[5]});
    h->kf_points += (size_t)nc + (size_t)ns;
    if (index_out) *index_out = (int32_t)h->keyPoses.size() - 1;
    return LVI_OK;
}
int32_t lvi_keyframe_add_current(lvi_lidar* h, const float pose[6], int32_t* index_out)
{
    if (!h || !pose) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->have_ds) return fail(LVI_ERR_STATE, "scan not downsampled");
    return lvi_keyframe_add(h, h->cornerDS.data(), (int32_t)h->cornerDS.size(), h->surfDS.data(), (int32_t)h->surfDS.size(), pose, index_out);
}
int32_t lvi_keyframe_set_pose(lvi_lidar* h, int32_t index, const float pose[6])
{
    if (!h || !pose || index < 0 || index >= (int32_t)h->keyPoses.size()) return fail(LVI_ERR_INVALID_ARG, "bad keyframe index");
    for (int k = 0; k < 6; k++) h->keyPoses[index][k] = pose[k];
    return LVI_OK;
}
int32_t lvi_keyframe_count(lvi_lidar* h, int32_t* n_keyframes, int32_t* n_points)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    if (n_keyframes) *n_keyframes = (int32_t)h->keyPoses.size();
    if (n_points) *n_points = (int32_t)h->kf_points;
    return LVI_OK;
}
int32_t lvi_keyframes_clear(lvi_lidar* h)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    h->cornerCloudKeyFrames.clear(); h->surfCloudKeyFrames.clear(); h->keyPoses.clear(); h->kf_points = 0;
    return LVI_OK;
}
int32_t lvi_map_assemble(lvi_lidar* h, const int32_t* key_indices, int32_t n_keys)
{
    if (!h || n_keys < 0 || (n_keys > 0 && !key_indices)) return fail(LVI_ERR_INVALID_ARG, "bad key list");
    size_t tc = 0, ts = 0;
    for (int i = 0; i < n_keys; i++) {
        const int k = key_indices[i];
        if (k < 0 || k >= (int)h->keyPoses.size()) return fail(LVI_ERR_INVALID_ARG, "key index out of range");
        tc += h->cornerCloudKeyFrames[k].size(); ts += h->surfCloudKeyFrames[k].size();
    }
    if (tc > (size_t)h->P.max_map_points || ts > (size_t)h->P.max_map_points) return fail(LVI_ERR_CAPACITY, "map exceeds max_map_points");
    std::vector<lvi_pt> c(tc), s(ts);
    size_t oc = 0, os = 0;
    for (int i = 0; i < n_keys; i++) {                                      // extractCloud :931-957 (cache or not: same points)
        const int k = key_indices[i];
        const lvo::Affine3f t = lvo::trans2Affine3f(h->keyPoses[k].data());    // transformPointCloud :347-354
        transformPointCloud(h->cornerCloudKeyFrames[k].data(), (int)h->cornerCloudKeyFrames[k].size(), t, c.data() + oc, h->P.numberOfCores);
        transformPointCloud(h->surfCloudKeyFrames[k].data(), (int)h->surfCloudKeyFrames[k].size(), t, s.data() + os, h->P.numberOfCores);
        oc += h->cornerCloudKeyFrames[k].size(); os += h->surfCloudKeyFrames[k].size();
    }
    return lvi_map_set(h, c.data(), (int32_t)tc, s.data(), (int32_t)ts);
}

// the reference has no incremental form: extractCloud fuses and re-filters the whole local map every scan (:931-965)
int32_t lvi_map_update(lvi_lidar* h, const int32_t* key_indices, int32_t n_keys) { return lvi_map_assemble(h, key_indices, n_keys); }

int32_t lvi_scan_set_deskew(lvi_lidar* h, const lvi_deskew_info* info)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    h->imu_available = false;
    if (!info || !info->imu_available) return LVI_OK;
    if (info->imu_pointer_cur < 1 || info->imu_pointer_cur >= LVI_DESKEW_MAX_IMU || !info->imu_time || !info->imu_rot_x || !info->imu_rot_y || !info->imu_rot_z)
        return fail(LVI_ERR_INVALID_ARG, "bad deskew table");
    const int m = info->imu_pointer_cur + 1;
    h->imuTime.assign(info->imu_time, info->imu_time + m);
    h->imuRotX.assign(info->imu_rot_x, info->imu_rot_x + m);
    h->imuRotY.assign(info->imu_rot_y, info->imu_rot_y + m);
    h->imuRotZ.assign(info->imu_rot_z, info->imu_rot_z + m);
    h->imuPointerCur = info->imu_pointer_cur; h->timeScanCur = info->time_scan_cur;
    h->imu_available = true;
    return LVI_OK;
}

// ---- one-call forms --------------------------------------------------------
int32_t lvi_organize_scan(lvi_lidar* h, const lvi_livox_pt* pts, int32_t n_raw, lvi_scan_info* out)
{
    int32_t st = lvi_scan_set_deskew(h, nullptr); if (st) return st;
    st = lvi_scan_upload(h, pts, n_raw); if (st) return st;
    st = lvi_scan_organize(h); if (st) return st;
    return lvi_get_scan_info(h, out);
}
int32_t lvi_organize_scan_deskew(lvi_lidar* h, const lvi_livox_pt* pts, int32_t n_raw, const lvi_deskew_info* info, lvi_scan_info* out)
{
    int32_t st = lvi_scan_set_deskew(h, info); if (st) return st;
    st = lvi_scan_upload(h, pts, n_raw); if (st) return st;
    st = lvi_scan_organize(h); if (st) return st;
    return lvi_get_scan_info(h, out);
}
int32_t lvi_extract_features(lvi_lidar* h, const lvi_scan_info* in, lvi_cloud* corner, lvi_cloud* surf)
{
    if (!h || !in) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (in->n < 0 || in->n > h->P.N_SCAN * h->P.Horizon_SCAN) return fail(LVI_ERR_CAPACITY, "scan_info.n out of range");
    h->extracted.assign(in->cloud_deskewed, in->cloud_deskewed + in->n);
    h->range.assign(in->point_range, in->point_range + in->n);
    h->col.assign(in->point_col_ind, in->point_col_ind + in->n);
    h->startR.assign(in->start_ring_index, in->start_ring_index + h->P.N_SCAN);
    h->endR.assign(in->end_ring_index, in->end_ring_index + h->P.N_SCAN);
    h->have_org = true;
    int32_t st = lvi_scan_extract(h); if (st) return st;
    return lvi_get_features(h, corner, surf);
}
int32_t lvi_voxel_downsample(lvi_lidar* h, const lvi_pt* in, int32_t n, float leaf, lvi_pt* out, int32_t out_capacity, int32_t* n_out)
{
    if (!h || n < 0 || (n > 0 && !in) || !(leaf > 0.f) || !n_out) return fail(LVI_ERR_INVALID_ARG, "bad voxel arguments");
    std::vector<lvi_pt> o;
    int m = lvo::voxel_grid_filter(in, n, leaf, o, &h->vdbg);
    *n_out = m;
    if (m > out_capacity) return fail(LVI_ERR_CAPACITY, "voxel output capacity too small");
    if (m) std::memcpy(out, o.data(), sizeof(lvi_pt) * m);
    return LVI_OK;
}
int32_t lvi_map_set(lvi_lidar* h, const lvi_pt* c, int32_t nc, const lvi_pt* s, int32_t ns)
{
    int32_t st = lvi_map_upload(h, c, nc, s, ns); if (st) return st;
    return lvi_map_build(h);
}
int32_t lvi_scan_to_map(lvi_lidar* h, const lvi_pt* corner, int32_t nc, const lvi_pt* surf, int32_t ns,
                        const lvi_imu_hint* imu, float pose[6], lvi_icp_result* out)
{
    if (!h || nc < 0 || ns < 0 || (nc > 0 && !corner) || (ns > 0 && !surf)) return fail(LVI_ERR_INVALID_ARG, "bad arguments");
    h->corner.assign(corner, corner + nc); h->surf.assign(surf, surf + ns);
    h->have_feat = true;
    downsampleCurrentScan(h);
    return lvi_scan_match(h, imu, pose, out);
}
int32_t lvi_transform_cloud(lvi_lidar* h, const lvi_pt* in, int32_t n, const float pose6[6], lvi_pt* out)
{
    if (!h || n < 0 || (n > 0 && (!in || !out)) || !pose6) return fail(LVI_ERR_INVALID_ARG, "bad arguments");
    transformPointCloud(in, n, lvo::trans2Affine3f(pose6), out, h->P.numberOfCores);
    return LVI_OK;
}

// ---- inspection -------------------------------------------------------------
int32_t lvi_debug_get(lvi_lidar* h, int32_t what, void* dst, int64_t cap, int64_t* n_bytes)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    const size_t n = h->extracted.size();
    switch (what) {
        case LVI_DBG_MAP_CORNER_RAW: if (!h->have_map_raw) break; return dbg_copy(h->mapCornerRaw.data(), h->mapCornerRaw.size(), dst, cap, n_bytes);
        case LVI_DBG_MAP_SURF_RAW:   if (!h->have_map_raw) break; return dbg_copy(h->mapSurfRaw.data(), h->mapSurfRaw.size(), dst, cap, n_bytes);
        case LVI_DBG_CURVATURE:    if (!h->have_feat) break; return dbg_copy(h->cloudCurvature.data(), n, dst, cap, n_bytes);
        case LVI_DBG_PICKED_OCCL:  if (!h->have_feat) break; return dbg_copy(h->pickedOccl.data(), n, dst, cap, n_bytes);
        case LVI_DBG_LABEL:        if (!h->have_feat) break; return dbg_copy(h->cloudLabel.data(), n, dst, cap, n_bytes);
        case LVI_DBG_PICKED_FINAL: if (!h->have_feat) break; return dbg_copy(h->cloudNeighborPicked.data(), n, dst, cap, n_bytes);
        case LVI_DBG_CORNER_INDEX: if (!h->have_feat) break; return dbg_copy(h->corner_index.data(), h->corner_index.size(), dst, cap, n_bytes);
        case LVI_DBG_VOXEL_KEYS:   return dbg_copy(h->vdbg.keys.data(), h->vdbg.keys.size(), dst, cap, n_bytes);
        case LVI_DBG_VOXEL_CELLS:  return dbg_copy(h->vdbg.cells.data(), h->vdbg.cells.size(), dst, cap, n_bytes);
        case LVI_DBG_VOXEL_COUNTS: return dbg_copy(h->vdbg.counts.data(), h->vdbg.counts.size(), dst, cap, n_bytes);
        case LVI_DBG_ICP_JTJ:      return dbg_copy(h->jtj_trace.data(), h->jtj_trace.size(), dst, cap, n_bytes);
        case LVI_DBG_ICP_POSE_TRACE: return dbg_copy(h->pose_trace.data(), h->pose_trace.size(), dst, cap, n_bytes);
        default: return fail(LVI_ERR_INVALID_ARG, "unknown debug item");
    }
    return fail(LVI_ERR_STATE, "stage not run");
}

int32_t lvi_debug_knn(lvi_lidar* h, int32_t which, const lvi_pt* queries, int32_t nq, int32_t* idx, float* sqd)
{
    if (!h || !queries || !idx || !sqd || nq < 0) return fail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (!h->have_map) return fail(LVI_ERR_STATE, "map not built");
    const lvo::KdTree3f& kd = which == 0 ? h->kdCorner : h->kdSurf;
    for (int i = 0; i < nq; i++) {
        const float q[3] = {queries[i].x, queries[i].y, queries[i].z};
        int ri[5]; float rd[5];
        int found = kd.knn(q, 5, ri, rd);
        for (int j = 0; j < 5; j++) {
            if (j < found && rd[j] < 1.0f) { idx[i * 5 + j] = ri[j]; sqd[i * 5 + j] = rd[j]; }
            else { idx[i * 5 + j] = -1; sqd[i * 5 + j] = INFINITY; }
        }
    }
    return LVI_OK;
}

int32_t lvi_debug_residuals(lvi_lidar* h, int32_t which, const float pose[6], lvi_pt* coeff, uint8_t* flag, int32_t capacity, int32_t* n)
{
    if (!h || !pose || !coeff || !flag || !n) return fail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (!h->have_map || !h->have_ds) return fail(LVI_ERR_STATE, "map or scan DS missing");
    const std::vector<lvi_pt>& q = which == 0 ? h->cornerDS : h->surfDS;
    *n = (int32_t)q.size();
    if (capacity < *n) return fail(LVI_ERR_CAPACITY, "capacity too small");
    const lvo::Affine3f T = lvo::trans2Affine3f(pose);
    for (int i = 0; i < *n; i++) {
        lvi_pt c = {0, 0, 0, 0};
        bool ok = which == 0 ? cornerResidual(h, T, q[i], c) : surfResidual(h, T, q[i], c);
        if (!ok) c = lvi_pt{0, 0, 0, 0};
        coeff[i] = c; flag[i] = ok ? 1 : 0;
    }
    return LVI_OK;
}

// hip-only entry points: present so that the symbol set is identical, but unsupported here
// ---- batched form: the reference processes scans one by one; a batch is that, n times ----------------------
int32_t lvi_scan_batch_upload(lvi_lidar* h, int32_t n_scans, const lvi_livox_pt* const* pts, const int32_t* n_raw)
{
    if (!h || !pts || !n_raw) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (n_scans < 1 || n_scans > std::max(h->P.batch_scans, 1)) return fail(LVI_ERR_CAPACITY, "n_scans exceeds lvi_lidar_params.batch_scans");
    for (int z = 0; z < n_scans; z++) {
        if (n_raw[z] < 0 || (n_raw[z] > 0 && !pts[z])) return fail(LVI_ERR_INVALID_ARG, "bad scan");
        if (n_raw[z] > h->P.max_raw_points) return fail(LVI_ERR_CAPACITY, "n_raw exceeds max_raw_points");
    }
    h->batch_scans.resize(n_scans);
    for (int z = 0; z < n_scans; z++) h->batch_scans[z].assign(pts[z], pts[z] + n_raw[z]);
    return LVI_OK;
}
// "device" memory of the CPU oracle is host memory
int32_t lvi_scan_batch_bind_device(lvi_lidar* h, int32_t n_scans, const void* const* d_pts, const int32_t* n_raw)
{
    return lvi_scan_batch_upload(h, n_scans, reinterpret_cast<const lvi_livox_pt* const*>(d_pts), n_raw);
}
int32_t lvi_scan_batch_run(lvi_lidar* h, int32_t n_scans, const float* pose_init, void* d_records, int32_t rebuild_map)
{
    if (!h || !pose_init) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (n_scans < 1 || n_scans > (int32_t)h->batch_scans.size()) return fail(LVI_ERR_STATE, "no scan bound / uploaded for a slot");
    if (rebuild_map && !h->have_map_raw) return fail(LVI_ERR_STATE, "no map uploaded");
    h->batch_records.assign(n_scans, lvi_pose_record{});
    for (int z = 0; z < n_scans; z++) {
        int32_t st = lvi_scan_upload(h, h->batch_scans[z].data(), (int32_t)h->batch_scans[z].size()); if (st < 0) return st;
        if (rebuild_map) { st = lvi_map_build(h); if (st < 0) return st; }
        st = lvi_scan_organize(h); if (st < 0) return st;
        st = lvi_scan_extract(h); if (st < 0) return st;
        st = lvi_scan_downsample(h); if (st < 0) return st;
        float pose[6]; lvi_icp_result res;
        std::memcpy(pose, pose_init + 6 * z, sizeof(pose));
        st = lvi_scan_match(h, nullptr, pose, &res); if (st < 0) return st;
        h->batch_records[z] = h->last_record;
        if (d_records) static_cast<lvi_pose_record*>(d_records)[z] = h->last_record;
    }
    return LVI_OK;
}
int32_t lvi_scan_batch_get_records(lvi_lidar* h, int32_t n_scans, lvi_pose_record* out)
{
    if (!h || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (n_scans < 1 || n_scans > (int32_t)h->batch_records.size()) return fail(LVI_ERR_STATE, "batch not run");
    std::copy(h->batch_records.begin(), h->batch_records.begin() + n_scans, out);
    return LVI_OK;
}
int32_t lvi_batch_select(lvi_lidar* h, int32_t slot) { return (h && slot == 0) ? LVI_OK : fail(LVI_ERR_UNSUPPORTED, "hip only: the oracle keeps the intermediates of the last scan only"); }

int32_t lvi_scan_match_async(lvi_lidar*, const float*, void*) { return fail(LVI_ERR_UNSUPPORTED, "hip only"); }
int32_t lvi_scan_upload_device(lvi_lidar* h, const void* d_pts, int32_t n_raw) { return lvi_scan_upload(h, static_cast<const lvi_livox_pt*>(d_pts), n_raw); }   // "device" memory of the CPU oracle is host memory
int32_t lvi_scan_replay_enqueue(lvi_lidar*, const void*, int32_t, const float*, void*, int32_t) { return fail(LVI_ERR_UNSUPPORTED, "hip only"); }
int32_t lvi_map_upload_device(lvi_lidar* h, const void* c, int32_t nc, const void* s, int32_t ns)
{
    return lvi_map_upload(h, static_cast<const lvi_pt*>(c), nc, static_cast<const lvi_pt*>(s), ns);          // "device" memory of the CPU oracle is host memory
}
// the HIP library aliases the owner's clouds; here they are copied (the same clouds, the same results)
int32_t lvi_map_share(lvi_lidar* h, lvi_lidar* owner)
{
    if (!h || !owner || h == owner) return fail(LVI_ERR_INVALID_ARG, "bad handles");
    if (!owner->have_map_raw) return fail(LVI_ERR_STATE, "the owner holds no map");
    if ((int)owner->mapCornerRaw.size() > h->P.max_map_points || (int)owner->mapSurfRaw.size() > h->P.max_map_points) return fail(LVI_ERR_CAPACITY, "map exceeds max_map_points");
    h->mapCornerRaw = owner->mapCornerRaw; h->mapSurfRaw = owner->mapSurfRaw;
    h->have_map_raw = true; h->have_map = false;
    return LVI_OK;
}
int32_t lvi_prof_enable(lvi_lidar*, int32_t) { return LVI_OK; }
int32_t lvi_prof_reset(lvi_lidar*) { return LVI_OK; }
int32_t lvi_prof_read(lvi_lidar*, lvi_kernel_stat*, int32_t, int32_t* n) { if (n) *n = 0; return LVI_OK; }

}  // extern "C"
