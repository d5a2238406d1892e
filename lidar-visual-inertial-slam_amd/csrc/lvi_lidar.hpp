// Device-resident state of one lidar handle and the stage launchers (implemented in
// lvi_scan.hip and lvi_icp.hip; lvi_capi.hip wires them to the C-ABI).
#pragma once
#include <array>

#include "lvi_voxel.hpp"

namespace lvi {

constexpr int ORG_TILE = 1024;          // raw points per workgroup in the organise kernels
constexpr int FEAT_SEG_CAP = 8192;      // max points of one ring sector held in LDS (+ halo)
constexpr int MAX_N_SCAN = 32;
constexpr int CORNERS_PER_SECTOR = 40;  // featureExtraction.cpp:180
constexpr int KNN_G = 8;                // lanes that share one KNN query (measured: 4 lanes 27 us / 1 880 scans/s, 8 lanes 24 us / 1 960, 16 lanes 30 us / 1 725)
constexpr int ICP_BLOCK = 64 * KNN_G;   // residual workgroup: 64 features x KNN_G lanes

// device status bits (sticky until the next upload)
enum { DEV_ERR_SECTOR_TOO_LARGE = 1, DEV_ERR_GRID_TOO_LARGE = 2, DEV_ERR_SECTOR_HANDOVER = 4 };

struct IcpPose {                         // written by icp_solve for the next residual pass
    float T[6];                          // transformTobeMapped: roll pitch yaw x y z
    float A[12];                         // trans2Affine3f(T), row-major 3x4
    float trig[6];                       // srx crx sry cry srz crz (LMOptimization :1202-1207)
};

struct IcpState {                        // device, one per handle
    IcpPose pose[2];                     // pose of Gauss-Newton iteration i in pose[i & 1] (a launch reads one buffer and writes the other)
    int cur;                             // buffer of the latest pose (transformUpdate reads it)
    IcpPose scratch;                     // lvi_transform_cloud / lvi_debug_residuals
    int done;                            // LMOptimization returned true (and break enabled) or loop skipped
    int converged, degenerate, iters, any_lm, status;
    int final_status;                    // what the scan's record says (status + LVI_TOO_FEW_CORRESPONDENCES / device errors), written by the finish step
    int n_sel[LVI_ICP_MAX_ITERS];
    float jtj[LVI_ICP_MAX_ITERS * 27];
    float pose_trace[(LVI_ICP_MAX_ITERS + 1) * 6];
    lvi_pose_record record;              // final
    float final_pose[6];
};

// what lvi_scan_match returns, written by the finish step into pinned host memory (no copy commands behind the loop: a 3 KB
// hipMemcpyAsync of the whole IcpState and two more for the counts and status words were three blit kernels per chunk)
struct IcpHostResult {
    int final_status, iters, converged, degenerate, done, status;
    int n_sel[LVI_ICP_MAX_ITERS];
    float final_pose[6];
    int nq[3];                           // voxScan.d_nout: corner_ds, surf_ds, total
    int dw[2];                           // device status words (scan side, map build)
};

struct GridIndex {                       // uniform 0.5 m grid over one DS map (a-6 replacement for the kd-tree)
    int* cell_start = nullptr;           // [max_cells + 2]
    int* count = nullptr;                // [max_cells + 2] points per cell (zero between builds)
    int* blockSum = nullptr;             // [1024] chunk totals of the cell scan
    lvi_pt* sorted = nullptr;            // [cap] xyz + original DS index in the intensity slot (as int bits)
    struct Meta { double origin[3]; double edge, inv_edge; int dim[3]; int ncells; int n; int ok; int R; }* meta = nullptr;   // device
};

struct LidarDev {
    lvi_lidar_params P;
    int device = 0;
    Ctx ctx;                                               // scan-side stages and the GN loop
    Ctx ctx2;                                              // map build (independent of the scan until scan matching)
    hipEvent_t evMain = nullptr, evMap = nullptr;
    hipEvent_t evMapDeps = nullptr; bool have_map_deps = false;   // main-stream work a later incremental map update (second stream) must wait for (mark_map_deps)
    hipEvent_t evMark[LVI_LIDAR_MARKS] = {};               // lvi_lidar_mark / lvi_lidar_wait_mark, created on first use
    bool map_pending = false;                              // map build enqueued on ctx2, not yet joined by ctx
    Profiler prof;
    Arena arena;

    // capacities
    int raw_cap = 0, ext_cap = 0, ring_cap = 0, map_cap = 0, nblk_org = 0, max_cells = 0;

    // ---- a-0
    lvi_livox_pt* raw = nullptr; int n_raw = 0;            // after dropping the last point
    lvi_livox_pt* h_raw[2] = {nullptr, nullptr}; hipEvent_t ev_raw[2] = {nullptr, nullptr}; int raw_slot = 0;   // pinned staging ring of lvi_scan_upload
    // ---- f-4 (keyframe store): clouds in the sensor frame, packed in one pool; tables on the host
    struct KfSeg { int in_off, n, out_off, which; float A[12]; };              // one (keyframe, corner|surf) piece of an assembly
    lvi_pt* kfPool = nullptr; int kf_pool_cap = 0, kf_pool_used = 0;
    std::vector<int> kf_off_c, kf_n_c, kf_off_s, kf_n_s;
    std::vector<std::array<float, 6>> kf_pose;
    KfSeg* d_kfSeg = nullptr; KfSeg* h_kfSeg = nullptr;                        // [2 * max_assemble_keys] device / pinned host
    int kf_seg_cap = 0;
    // ---- f-1 (IMU deskew): imuDeskewInfo's table, set by lvi_scan_set_deskew
    bool dk_on = false; int dk_cur = 0; double dk_t0 = 0.0;
    double* d_dk = nullptr;                                // [4][LVI_DESKEW_MAX_IMU] imuTime, imuRotX, imuRotY, imuRotZ
    int* d_dk_first = nullptr;                             // message index of the first point that reaches deskewPoint (INT_MAX between scans)
    float* d_dk_startInv = nullptr;                        // [9] transStartInverse, linear part
    int* blockCnt = nullptr;                               // [N_SCAN][nblk_org]
    int* ringBase = nullptr;                               // [N_SCAN + 1] (kept columns)
    int *startR = nullptr, *endR = nullptr, *d_n = nullptr;
    lvi_pt* pts = nullptr; float* range = nullptr; int* col = nullptr;
    // ---- a-1..a-3
    float* curv = nullptr; uint8_t *picked = nullptr, *picked_occl = nullptr, *surfmask = nullptr, *pflags = nullptr; int8_t* label = nullptr;
    unsigned* sectorSpill = nullptr;                       // [MAX_N_SCAN * 6][2] hand-over words of the pipelined sector kernel
    long long feat_handover_ticks = 200000;                // 2 ms at 100 MHz: wait of a pipelined sector workgroup before it redoes the ring itself (LVI_FEAT_HANDOVER_TICKS)
    int *sector_idx = nullptr, *sector_cnt = nullptr;      // [N_SCAN*6*40], [N_SCAN*6]
    lvi_pt* corner = nullptr; int* corner_idx = nullptr; int* d_ncorner = nullptr;
    lvi_pt* surf = nullptr;                                // concatenated per-ring DS output
    int* d_fresh = nullptr;                                // 1 until the first extract of this handle
    int* d_status = nullptr;
    long long* d_feat_cycles = nullptr;                    // [8] phase cycle counters of ring 0 (diagnostics)
    long long* d_icp_cycles = nullptr;                     // [16] phase cycle counters: [0..7] residual workgroup 0, [8..15] solve kernel
    VoxelPlan voxRing;                                     // N_SCAN segments, leaf odometrySurfLeafSize
    // ---- scan DS
    lvi_pt *cornerDS = nullptr, *surfDS = nullptr;
    VoxelPlan voxScan;                                     // 2 segments (corner, surf)
    // ---- map
    lvi_pt *mapCornerRaw = nullptr, *mapSurfRaw = nullptr, *mapCornerDS = nullptr, *mapSurfDS = nullptr;
    lvi_pt *mapCornerOwn = nullptr, *mapSurfOwn = nullptr;     // slot 0: the handle's own raw-map memory (mapCornerRaw / mapSurfRaw point elsewhere while lvi_map_share is in force)
    int n_map_corner = 0, n_map_surf = 0;
    VoxelPlan voxMap;                                      // 2 segments
    GridIndex grid[2];

    // ---- generic one-call voxel (lvi_voxel_downsample)
    lvi_pt *genIn = nullptr, *genOut = nullptr;
    VoxelPlan voxGen;                                      // 1 segment
    unsigned* genKeysDbg = nullptr;
    // ---- icp
    IcpState* icp = nullptr;
    unsigned long long* icpAcc = nullptr;                  // [3][8][56] exact fixed-point totals of the 28 sums of a GN launch (coarse, fine): 8 shards, 3 buffers in rotation
    lvi_pt* coeff = nullptr; uint8_t* flag = nullptr;      // [ext_cap] lvi_debug_residuals only
    int* nnPrev = nullptr;                                 // [5][ext_cap] the five neighbours of the feature's last search, in its fit's order
    float4* nnPt = nullptr;                                // [5][ext_cap] their coordinates
    float4 *fitA = nullptr, *fitB = nullptr;               // [ext_cap] the line / plane fitted to them
    unsigned char* fitOk = nullptr;                        // [ext_cap]
    int icp_g0 = 4;                                        // lanes per feature in GN iteration 0 (whole unit ball; LVI_ICP_G0; measured with 16 scans in flight: 8 lanes 5 030, 4 lanes 5 245 scans/s)
    int icp_g1 = 4;                                        // lanes per feature in GN iterations >= 1 (LVI_ICP_G1 = 8 | 4 | 2 | 84 (8 lanes, batches of 4))
    bool knn_tiles = false;                                // LVI_KNN_TILES=1 at create: a wavefront's searches read an LDS tile of the index when its features' reach fits one (tests: same bits)
    int icp_stamp_iter = -1;                               // LVI_ICP_STAMP_ITER: the GN iteration whose phase stamps LVI_DBG_ICP_CYCLES shows (-1: the last launched)
    int icp_wide_from = 3;                                 // first GN iteration that runs 256 features per workgroup (LVI_ICP_WIDE_FROM)
    float4* nnRef = nullptr;                               // [ext_cap] position at the feature's last search + squared lower bound on the distance to the map points outside its five
    bool knn_skip = true;                                  // LVI_KNN_NO_SKIP=1 at create: every iteration >= 1 runs its (bounded) search (tests: same bits)
    float knn_slack = 0.05f;                               // LVI_KNN_SLACK (m): radius added to a bounded search so that later iterations can skip theirs
    bool knn_bound = true;                                 // LVI_KNN_NO_BOUND=1 at create: every iteration searches the whole unit ball (tests: same bits)
    int nblk_icp = 0;
    IcpState* h_icp = nullptr;                             // pinned host mirror
    IcpHostResult* h_res = nullptr;                        // pinned: the finish step's result block
    int* h_gn_feat = nullptr;                              // pinned: features (corner + surf) of the last finished scan match of this slot, written by icp_final
                                                           // (a HINT for the next GN launches' grid: read without synchronising, any value is correct)
    float* d_pose_init = nullptr;                          // [6] initial guess of the next scan match (device)
    // captured launch sequence of the whole per-scan path (lvi_scan_replay_enqueue)
    hipGraphExec_t graphExec = nullptr;
    std::array<int, 8> graph_key = {-1, -1, -1, -1, -1, -1, -1, -1};   // what the capture froze (lvi_scan_replay_enqueue)
    // stage flags (host)
    bool have_raw = false, have_org = false, have_feat = false, have_ds = false, have_map_raw = false, have_map = false;
    bool gen_valid = false; int gen_n = 0;
    bool gen_static_set = false; float gen_leaf = 0.f;     // lvi_voxel_downsample: the segment table on the device is for this leaf size
    // ---- f-4: incremental local map (lvi_map_update); slot 0 of a non-batch handle only
    IncMap inc;
    bool inc_ready = false;                                // tables hold exactly inc_mult's keyframes at inc_pose's poses
    std::vector<int> inc_mult;                             // [n keyframes] multiplicity of the key in the current list
    std::vector<std::array<float, 6>> inc_pose;            // pose each active key was added with
    int inc_nocc_bound = 0;                                // upper bound of the occupied slots (tombstones included)
    // batch slots (lvi_lidar_params.batch_scans > 1): slot z > 0 shares slot 0's streams, profiler, keyframe store and RAW
    // local map (the replay configuration: one frozen raw map, re-voxelised and re-indexed for every scan)
    LidarDev* map_owner = nullptr;                         // slot 0 for z > 0
    const lvi_livox_pt* raw_bound = nullptr;               // lvi_scan_batch_bind_device: the scan is read in place (no copy)
};

// the scans of one batched launch sequence: slot 0 owns the streams, the profiler and the raw local map
struct Slots {
    LidarDev* const* p; int n;
    LidarDev& operator[](int i) const { return *p[i]; }
    LidarDev& first() const { return *p[0]; }
};
struct OneSlot { LidarDev* one; Slots s; explicit OneSlot(LidarDev& d) : one(&d), s{&one, 1} {} };

// lvi_scan.hip
void lidar_allocate(LidarDev& d);
void stage_organize(const Slots& s);
void stage_extract(const Slots& s);
void stage_downsample(const Slots& s);
void stage_map_build(const Slots& s);
// Gauss-Newton iterations [it_begin, it_end) (it_end < 0: to icp_max_iters) followed by the finish step; slot z writes its record to d_records + 32 z (when not null)
void stage_scan_match_enqueue(const Slots& s, const lvi_imu_hint* imu, void* d_records, int it_begin = 0, int it_end = -1);
void set_pose_init(const Slots& s, const float* pose_init, bool clear_status);          // [n][6]; clear_status: also zero the scan-side device status words
bool stage_map_update(LidarDev& d, const int32_t* keys, int n_keys);       // f-4 incremental: false = not applicable / device said no → the caller assembles
void stage_map_index(const Slots& s, const Ctx& cx);                     // the KNN grid index over every slot's DS map
void stage_map_assemble(LidarDev& d, const int32_t* keys, int n_keys);      // f-4: fuse the keyframes into the raw map buffers (the caller builds)
void stage_organize(LidarDev& d);
void stage_extract(LidarDev& d);
void stage_downsample(LidarDev& d);
// lvi_icp.hip
void stage_map_build(LidarDev& d);
void join_map(LidarDev& d);                                // make the main stream wait for a pending map build
void mark_map_deps(LidarDev& d);                           // record: a later map update on the second stream waits for everything enqueued on the main stream so far
void set_pose_init(LidarDev& d, const float pose_init[6]);  // enqueue: d_pose_init <- pose_init
void stage_scan_match_enqueue(LidarDev& d, const lvi_imu_hint* imu, void* d_record, int it_begin = 0, int it_end = -1);   // starts from d_pose_init
void debug_knn(LidarDev& d, int which, const lvi_pt* d_queries, int nq, int* d_idx, float* d_sqd);
void debug_residuals(LidarDev& d, int which, const float pose[6]);
void transform_cloud(LidarDev& d, const lvi_pt* d_in, int n, const float pose6[6], lvi_pt* d_out);

}  // namespace lvi
