// C entry points over the host mirror (lvi_host.hpp) for replay harnesses that are not C++: the sequential
// lidar_odometry loop (imageProjection → featureExtraction → mapOptimization, one scan after the other, every scan
// matched against the local map its predecessors built) and the feature_tracker node callback.  The logic lives in
// lvi_host.hpp; this file only flattens it to plain C.  Links against any library exporting include/lvi_hotpath.h:
// liblvi_hip.so in deployment (build.py → host/liblvi_host_hip.so), the CPU oracle in the CPU-tier tests.
#include <cstring>
#include <memory>

#include "lvi_host.hpp"

using namespace lvi_host;

namespace {
thread_local std::string g_err;
template <class F> int32_t guarded(F&& f)
{
    try { return f(); }
    catch (const Error& e) { g_err = e.what(); return e.code; }
    catch (const std::exception& e) { g_err = e.what(); return LVI_ERR_HIP; }
}
}  // namespace

struct lvh_seq {
    std::unique_ptr<LidarHandle> h;
    std::unique_ptr<MapOptimizationNode> mo;
};

struct lvh_trk {
    std::unique_ptr<TrackerHandle> t;
    std::unique_ptr<FeatureTracker> ft;
    std::unique_ptr<FeatureTrackerNode> node;
};

extern "C" {

typedef struct lvh_seq_params {
    int32_t incremental_map;                    // lvi_map_update instead of lvi_map_assemble
    int32_t use_imu_heading_initialization;
    double  mapping_process_interval;
    float   keyframe_adding_dist, keyframe_adding_angle, keyframe_density, keyframe_search_radius;
} lvh_seq_params;

typedef struct lvh_seq_result {
    int32_t processed;                          // 0: dropped by the mappingProcessInterval gate
    int32_t status, iters, converged, degenerate;
    int32_t saved_keyframe, n_keyframes, n_keys;
    float   pose[6];                            // transformTobeMapped after the scan
} lvh_seq_result;

const char* lvh_last_error(void) { return g_err.c_str(); }

void lvh_seq_params_default(lvh_seq_params* p)
{
    const MapCallerParams d;
    p->incremental_map = 1; p->use_imu_heading_initialization = d.useImuHeadingInitialization ? 1 : 0;
    p->mapping_process_interval = d.mappingProcessInterval;
    p->keyframe_adding_dist = d.surroundingkeyframeAddingDistThreshold; p->keyframe_adding_angle = d.surroundingkeyframeAddingAngleThreshold;
    p->keyframe_density = d.surroundingKeyframeDensity; p->keyframe_search_radius = d.surroundingKeyframeSearchRadius;
}

lvh_seq* lvh_seq_create(const lvi_lidar_params* lp, int32_t device, const lvh_seq_params* sp)
{
    lvh_seq* s = new lvh_seq();
    const int32_t st = guarded([&]() -> int32_t {
        s->h.reset(new LidarHandle(*lp, device));
        MapCallerParams P;
        if (sp) {
            P.incrementalMap = sp->incremental_map != 0; P.useImuHeadingInitialization = sp->use_imu_heading_initialization != 0;
            P.mappingProcessInterval = sp->mapping_process_interval;
            P.surroundingkeyframeAddingDistThreshold = sp->keyframe_adding_dist; P.surroundingkeyframeAddingAngleThreshold = sp->keyframe_adding_angle;
            P.surroundingKeyframeDensity = sp->keyframe_density; P.surroundingKeyframeSearchRadius = sp->keyframe_search_radius;
        }
        s->mo.reset(new MapOptimizationNode(*s->h, P));
        return LVI_OK;
    });
    if (st != LVI_OK) { delete s; return nullptr; }
    return s;
}
void lvh_seq_destroy(lvh_seq* s) { delete s; }
lvi_lidar* lvh_seq_handle(lvh_seq* s) { return s ? s->h->get() : nullptr; }

static void fill_result(lvh_seq* s, bool processed, lvh_seq_result* out)
{
    if (!out) return;
    std::memset(out, 0, sizeof(*out));
    const MapOptimizationNode& m = *s->mo;
    out->processed = processed ? 1 : 0;
    out->status = m.lastStatus; out->iters = m.last.iters; out->converged = m.last.converged; out->degenerate = m.last.degenerate;
    out->saved_keyframe = m.lastSavedKeyFrame ? 1 : 0; out->n_keyframes = (int32_t)m.cloudKeyPoses3D.size(); out->n_keys = (int32_t)m.lastKeys.size();
    for (int k = 0; k < 6; k++) out->pose[k] = m.transformTobeMapped[k];
}

// one livox message through the three nodes; the feature clouds stay on the device between featureExtraction and
// mapOptimization (in the ROS graph they travel as a CloudInfo message — host/ros2/*.cpp)
int32_t lvh_seq_scan(lvh_seq* s, const lvi_livox_pt* pts, int32_t n_raw, double stamp, int32_t imu_available, float imu_roll, float imu_pitch, float imu_yaw,
                     lvh_seq_result* out)
{
    if (!s || (n_raw > 0 && !pts)) { g_err = "null argument"; return LVI_ERR_INVALID_ARG; }
    return guarded([&]() -> int32_t {
        lvi_lidar* h = s->h->get();
        check(lvi_scan_upload(h, pts, n_raw), "lvi_scan_upload");
        check(lvi_scan_organize(h), "lvi_scan_organize");
        check(lvi_scan_extract(h), "lvi_scan_extract");
        const lvi_imu_hint imu{imu_available, imu_roll, imu_pitch, imu_yaw};
        const bool done = s->mo->processResidentScan(stamp, imu);
        fill_result(s, done, out);
        return LVI_OK;
    });
}
// the same with the scan already in device memory (hip backend)
int32_t lvh_seq_scan_device(lvh_seq* s, const void* d_pts, int32_t n_raw, double stamp, int32_t imu_available, float imu_roll, float imu_pitch, float imu_yaw,
                            lvh_seq_result* out)
{
    if (!s || (n_raw > 0 && !d_pts)) { g_err = "null argument"; return LVI_ERR_INVALID_ARG; }
    return guarded([&]() -> int32_t {
        lvi_lidar* h = s->h->get();
        check(lvi_scan_upload_device(h, d_pts, n_raw), "lvi_scan_upload_device");
        check(lvi_scan_organize(h), "lvi_scan_organize");
        check(lvi_scan_extract(h), "lvi_scan_extract");
        const lvi_imu_hint imu{imu_available, imu_roll, imu_pitch, imu_yaw};
        const bool done = s->mo->processResidentScan(stamp, imu);
        fill_result(s, done, out);
        return LVI_OK;
    });
}
// a keyframe of an earlier session; the node's pose becomes the seeded one
int32_t lvh_seq_seed_keyframe(lvh_seq* s, const lvi_pt* corner, int32_t nc, const lvi_pt* surf, int32_t ns, const float pose[6], double time)
{
    if (!s || !pose || (nc > 0 && !corner) || (ns > 0 && !surf)) { g_err = "null argument"; return LVI_ERR_INVALID_ARG; }
    return guarded([&]() -> int32_t { return s->mo->seedKeyFrame(corner, nc, surf, ns, pose, time); });
}
// key indices of the last extractCloud, in fuse order
int32_t lvh_seq_keys(lvh_seq* s, int32_t* keys, int32_t capacity, int32_t* n)
{
    if (!s || !n) { g_err = "null argument"; return LVI_ERR_INVALID_ARG; }
    *n = (int32_t)s->mo->lastKeys.size();
    if (keys) { if (capacity < *n) { g_err = "capacity too small"; return LVI_ERR_CAPACITY; } std::memcpy(keys, s->mo->lastKeys.data(), sizeof(int32_t) * (size_t)*n); }
    return LVI_OK;
}
// cloudKeyPoses6D: [n][8] = x y z roll pitch yaw time index
int32_t lvh_seq_keyposes(lvh_seq* s, double* rows, int32_t capacity, int32_t* n)
{
    if (!s || !n) { g_err = "null argument"; return LVI_ERR_INVALID_ARG; }
    *n = (int32_t)s->mo->cloudKeyPoses6D.size();
    if (rows) {
        if (capacity < *n) { g_err = "capacity too small"; return LVI_ERR_CAPACITY; }
        for (int i = 0; i < *n; i++) {
            const PointTypePose& p = s->mo->cloudKeyPoses6D[i];
            const double r[8] = {p.x, p.y, p.z, p.roll, p.pitch, p.yaw, p.time, p.intensity};
            std::memcpy(rows + 8 * (size_t)i, r, sizeof(r));
        }
    }
    return LVI_OK;
}

// ---------------------------------------------------------------------------------------------- feature_tracker node
typedef void (*lvh_fundamental_fn)(const float* un_cur_xy, const float* un_forw_xy, int32_t n, double f_threshold, uint8_t* status, void* user);

lvh_trk* lvh_trk_create(const lvi_tracker_params* tp, int32_t device, int32_t row, int32_t col, int32_t freq, int32_t equalize, const lvi_mei_params* cam)
{
    lvh_trk* t = new lvh_trk();
    const int32_t st = guarded([&]() -> int32_t {
        t->t.reset(new TrackerHandle(*tp, device));
        t->ft.reset(new FeatureTracker(*t->t, row, col, tp->max_cnt, (int)tp->min_dist));
        if (equalize) t->ft->setEqualize(true);
        if (cam) t->ft->setCamera(*cam);
        t->node.reset(new FeatureTrackerNode(*t->ft, freq));
        return LVI_OK;
    });
    if (st != LVI_OK) { delete t; return nullptr; }
    return t;
}
void lvh_trk_destroy(lvh_trk* t) { delete t; }

void lvh_trk_set_fundamental_hook(lvh_trk* t, lvh_fundamental_fn fn, void* user)
{
    if (!t) return;
    if (!fn) { t->ft->findFundamentalMat = nullptr; return; }
    t->ft->findFundamentalMat = [fn, user](const std::vector<Point2f>& a, const std::vector<Point2f>& b, double thr, std::vector<uint8_t>& status) {
        fn(a.empty() ? nullptr : &a[0].x, b.empty() ? nullptr : &b[0].x, (int32_t)a.size(), thr, status.data(), user);
    };
}

// img_callback.  outcome: FeatureTrackerNode::Outcome.  When a message was assembled (outcome 3 or 4): n_points, points
// [n][3], channels [6][capacity] (id, u, v, vx, vy, depth).  pub_this_frame / rejectWithF_skipped / n_tracked for tests.
int32_t lvh_trk_image(lvh_trk* t, const uint8_t* img, double stamp, int32_t* outcome, int32_t* n_points, float* points_xyz, float* channels, int32_t capacity,
                      int32_t* info /* [4]: PUB_THIS_FRAME, rejectWithF_skipped, cur_pts, pub_count */)
{
    if (!t || !img || !outcome || !n_points) { g_err = "null argument"; return LVI_ERR_INVALID_ARG; }
    return guarded([&]() -> int32_t {
        FeatureMsg msg;
        const auto oc = t->node->img_callback(img, stamp, &msg);
        *outcome = (int32_t)oc;
        *n_points = (oc == FeatureTrackerNode::FIRST_PUBLISH_SUPPRESSED || oc == FeatureTrackerNode::PUBLISHED) ? (int32_t)msg.points.size() : 0;
        if (*n_points > capacity) { g_err = "capacity too small"; return LVI_ERR_CAPACITY; }
        for (int i = 0; i < *n_points; i++) {
            if (points_xyz) { points_xyz[3 * i] = msg.points[i].x; points_xyz[3 * i + 1] = msg.points[i].y; points_xyz[3 * i + 2] = msg.points[i].z; }
            if (channels) for (int c = 0; c < 6; c++) channels[(size_t)c * capacity + i] = msg.channels[c][i];
        }
        if (info) { info[0] = t->ft->PUB_THIS_FRAME ? 1 : 0; info[1] = t->ft->rejectWithF_skipped; info[2] = (int32_t)t->ft->cur_pts.size(); info[3] = t->node->pub_count; }
        return LVI_OK;
    });
}
// cur_pts / ids / track_cnt of the tracker after the last callback: [n][4] = x, y, id, track_cnt
int32_t lvh_trk_points(lvh_trk* t, float* rows, int32_t capacity, int32_t* n)
{
    if (!t || !n) { g_err = "null argument"; return LVI_ERR_INVALID_ARG; }
    *n = (int32_t)t->ft->cur_pts.size();
    if (rows) {
        if (capacity < *n) { g_err = "capacity too small"; return LVI_ERR_CAPACITY; }
        for (int i = 0; i < *n; i++) { rows[4 * i] = t->ft->cur_pts[i].x; rows[4 * i + 1] = t->ft->cur_pts[i].y; rows[4 * i + 2] = (float)t->ft->ids[i]; rows[4 * i + 3] = (float)t->ft->track_cnt[i]; }
    }
    return LVI_OK;
}

}  // extern "C"
