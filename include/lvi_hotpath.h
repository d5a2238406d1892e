/*
 * lvi_hotpath.h — C-ABI of the MI355X-native hot path of
 * valentinomario/LiDAR-Visual-Inertial-SLAM.
 *
 * The reference has no FFI / plugin interface of its own: the algorithms are
 * member functions of ROS 2 node classes.  This header is the seam a maintainer
 * binds instead of those member functions; every entry point names the
 * reference call site it replaces (paths relative to the reference tree,
 * lidar_odometry/src/… and feature_tracker/src/…).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch/ROS types cross the ABI.
 *   - every function returns an int32 status: 0 OK, <0 hard error,
 *     >0 soft outcome mirrored from the reference (it warned / skipped / returned false).
 *   - the caller owns every host buffer; the library owns device memory behind
 *     the opaque handles.  No exception crosses the ABI.
 *   - one handle per node; calls on one handle are serialised by the caller
 *     (SingleThreadedExecutor featureExtraction.cpp:273, lock_guard
 *     mapOptimization.cpp:309).  Handles on different GPUs are independent.
 *   - two libraries export these symbols: liblvi_hip.so (the product, HIP/gfx950)
 *     and oracle/liblvi_oracle.so (CPU restatement, test infrastructure only).
 *     Entry points marked [hip only] exist only in the product library.
 */
#ifndef LVI_HOTPATH_H
#define LVI_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVI_ABI_VERSION 6

/* ---- status codes -------------------------------------------------------- */
#define LVI_OK                        0
#define LVI_ERR_INVALID_ARG          -1
#define LVI_ERR_NO_DEVICE            -2   /* no HIP device / HIP runtime failure at create */
#define LVI_ERR_HIP                  -3   /* a HIP call failed; see lvi_last_error() */
#define LVI_ERR_CAPACITY             -4   /* input exceeds the capacity given at create */
#define LVI_ERR_STATE                -5   /* stage called before its producer stage */
#define LVI_ERR_UNSUPPORTED          -6
/* soft outcomes (reference behaviour, not failures) */
#define LVI_TOO_FEW_FEATURES          1   /* mapOptimization.cpp:1320,1341  "Not enough features!" */
#define LVI_TOO_FEW_CORRESPONDENCES   2   /* mapOptimization.cpp:1210  LMOptimization returned false on <50 rows */
#define LVI_NO_MAP                    3   /* mapOptimization.cpp:1317  no key poses yet → no scan matching */

/* ---- plain data ---------------------------------------------------------- */

/* pcl::PointXYZI as used on the path (utility.h:64).  PCL stores 32 B/point; only
 * these 16 B carry information and this is the layout used in HBM. */
typedef struct lvi_pt { float x, y, z, intensity; } lvi_pt;

/* livox_ros_driver2/CustomPoint (imageProjection.cpp:17-29, 249-258). */
typedef struct lvi_livox_pt {
    float    x, y, z;
    uint8_t  reflectivity;
    uint8_t  tag;
    uint8_t  line;
    uint8_t  _pad;
    uint32_t offset_time;     /* ns from the scan header stamp */
} lvi_livox_pt;

/* ParamServer fields read on the path (utility.h:156-313, params_lidar.yaml). */
typedef struct lvi_lidar_params {
    int32_t N_SCAN;                     /* yaml 4 */
    int32_t Horizon_SCAN;               /* yaml 6000 */
    int32_t downsampleRate;             /* yaml 1 */
    float   lidarMinRange;              /* yaml 1.0 */
    float   lidarMaxRange;              /* yaml 100.0 */
    float   edgeThreshold;              /* yaml 1.0 */
    float   surfThreshold;              /* yaml 0.1 */
    int32_t edgeFeatureMinValidNum;     /* yaml 10 */
    int32_t surfFeatureMinValidNum;     /* yaml 100 */
    float   odometrySurfLeafSize;       /* yaml 0.4 */
    float   mappingCornerLeafSize;      /* yaml 0.2 */
    float   mappingSurfLeafSize;        /* yaml 0.4 */
    float   z_tollerance;               /* yaml 1000 */
    float   rotation_tollerance;        /* yaml 1000 */
    float   imuRPYWeight;               /* yaml 0.01 */
    int32_t numberOfCores;              /* yaml 8; OpenMP threads of the CPU oracle only */
    /* scan2MapOptimization loop (mapOptimization.cpp:1325-1337) */
    int32_t icp_max_iters;              /* 20 */
    int32_t icp_disable_break;          /* 0 = reference semantics; 1 = run exactly icp_max_iters (throughput runs) */
    /* capacities (sizes of the device arenas) */
    int32_t max_raw_points;             /* per scan, Msg.point_num upper bound */
    int32_t max_map_points;             /* raw local-map points, corner + surf each (<= 2^25) */
    /* realisation of the voxel grids: 0 = auto (per batch, from the previous batch's grid size), 1 = sorted, 2 = binned.
     * Same output bits either way; only the HIP backend reads it. */
    int32_t voxel_mode;
    /* f-4: device-resident keyframe store (cornerCloudKeyFrames / surfCloudKeyFrames); 0 keyframes = no store */
    int32_t max_keyframes;              /* default 1024 */
    int32_t max_keyframe_points;        /* corner + surf points of all stored keyframes together, default 2^22 */
    /* 0 (default): the map build runs on a second stream of the handle and overlaps the scan-side stages — best for one
     * scan in flight (0.88 vs 1.00 ms).  1: everything on the handle's main stream — best when >= 4 handles are in flight,
     * one stream per hardware queue (HIP's default is 4): 2 150 vs 2 030 scans/s.  Only the HIP backend reads it. */
    int32_t map_on_main_stream;
    /* The sector kernel runs the six sectors of a ring as a pipeline of workgroups; a workgroup that has not received its
     * predecessor's hand-over word after this many microseconds walks the ring from sector 0 itself (same results, no
     * workgroup ever depends on another one being resident).  0 = default (2000), < 0 = never wait (tests).  HIP backend only. */
    int32_t sector_handover_wait_us;
    /* Capacity of the batched entry points (lvi_scan_batch_*): scans processed side by side by ONE launch sequence, every
     * kernel carrying the scan index in blockIdx.z.  1 (default) = no batch slots beyond the handle's own. <= LVI_MAX_BATCH. */
    int32_t batch_scans;
    /* Re-voxelisation of the raw local map (lvi_map_build, lvi_scan_batch_run / lvi_scan_replay_enqueue with rebuild_map, lvi_map_assemble).
     * 0 (default): every rebuild computes what VoxelGrid::filter computes for every scan (mapOptimization.cpp:958-965): the bounding box
     * (getMinMax3D) and the per-bin point counts, then partition and centroids — per batch slot.  1: bounding box, per-bin counts and partition
     * offsets are computed once, when the raw map is written (upload / assembly), and re-used by every rebuild of the unchanged map — the
     * same output bits, less work than the reference does; for replay against a frozen raw map.  Only the HIP backend reads it. */
    int32_t map_plan_cache;
} lvi_lidar_params;
#define LVI_MAX_BATCH 8

/* CloudInfo.msg:4-8 arrays + cloud_deskewed, as plain caller-owned arrays.
 * capacity = number of elements the three per-point arrays can hold. */
typedef struct lvi_scan_info {
    int32_t  capacity;
    int32_t  n;                  /* out: extractedCloud->size() */
    int32_t *start_ring_index;   /* [N_SCAN] */
    int32_t *end_ring_index;     /* [N_SCAN] */
    int32_t *point_col_ind;      /* [capacity] */
    float   *point_range;        /* [capacity] */
    lvi_pt  *cloud_deskewed;     /* [capacity] */
} lvi_scan_info;

/* caller-owned point array with capacity; n is written by the library */
typedef struct lvi_cloud {
    int32_t capacity;
    int32_t n;
    lvi_pt *pts;
} lvi_cloud;

/* The rotation table imuDeskewInfo() integrates from the IMU queue (imageProjection.cpp:354-410), as plain
 * caller-owned arrays: the queue handling stays in the node, the per-point work (findRotation :495-520,
 * deskewPoint :538-568) moves into lvi_scan_organize.  SURVEY §8 row f-1. */
#define LVI_DESKEW_MAX_IMU 2000     /* queueLength, imageProjection.cpp:50 */
typedef struct lvi_deskew_info {
    int32_t imu_available;          /* cloudInfo.imu_available (:408); 0 = deskewPoint returns the point */
    int32_t imu_pointer_cur;        /* imuPointerCur after :404 = last valid table index, >= 1 when available */
    double  time_scan_cur;          /* timeScanCur (:283), seconds */
    const double *imu_time;         /* [imu_pointer_cur + 1]  imuTime[] */
    const double *imu_rot_x;        /* [imu_pointer_cur + 1]  imuRotX[] … */
    const double *imu_rot_y;
    const double *imu_rot_z;
} lvi_deskew_info;

/* CloudInfo fields consumed by transformUpdate (mapOptimization.cpp:1345-1368) */
typedef struct lvi_imu_hint {
    int32_t imu_available;
    float   imu_roll_init, imu_pitch_init, imu_yaw_init;
} lvi_imu_hint;

#define LVI_ICP_MAX_ITERS 64
typedef struct lvi_icp_result {
    int32_t status;              /* LVI_OK or a soft outcome */
    int32_t iters;               /* iterations executed (LMOptimization calls) */
    int32_t converged;           /* LMOptimization returned true */
    int32_t degenerate;          /* isDegenerate (mapOptimization.cpp:1271-1283) */
    int32_t n_corner_ds, n_surf_ds;          /* laserCloud{Corner,Surf}LastDSNum */
    int32_t n_sel[LVI_ICP_MAX_ITERS];        /* laserCloudSelNum per iteration */
    float   pose[6];             /* transformTobeMapped after transformUpdate: roll,pitch,yaw,x,y,z */
} lvi_icp_result;

/* the 32-byte record gathered across GPUs (one per scan) */
typedef struct lvi_pose_record {
    float   pose[6];
    int32_t status;
    int32_t iters;
} lvi_pose_record;

typedef struct lvi_lidar lvi_lidar;        /* opaque */

/* ---- library ------------------------------------------------------------- */
int32_t     lvi_abi_version(void);
const char *lvi_backend(void);             /* "hip-gfx950" or "cpu-oracle" */
const char *lvi_last_error(void);          /* thread-local text of the last hard error */

/* ---- lidar handle -------------------------------------------------------- */
void    lvi_lidar_params_default(lvi_lidar_params *p);      /* values of params_lidar.yaml */
int32_t lvi_lidar_create(const lvi_lidar_params *p, int32_t device, lvi_lidar **out);
void    lvi_lidar_destroy(lvi_lidar *h);
int32_t lvi_lidar_sync(lvi_lidar *h);                       /* wait for the handle's stream */
/* queue depth > 1: lvi_lidar_mark records a point in the handle's stream order (after everything enqueued so far),
 * lvi_lidar_wait_mark blocks the caller until that point has been reached; a slot never marked returns at once.
 * A replayer keeps D scans enqueued per handle by marking slot i % D after scan i and waiting on it before scan i + D. */
#define LVI_LIDAR_MARKS 8
int32_t lvi_lidar_mark(lvi_lidar *h, int32_t slot);
int32_t lvi_lidar_wait_mark(lvi_lidar *h, int32_t slot);

/* ---- one-call entry points with host buffers (a maintainer's drop-in seams) */

/* a-0  ImageProjection::moveFromCustomMsg + projectPointCloud + cloudExtraction
 *      (imageProjection.cpp:239-260, 570-647), imu_available == false (no deskew).
 *      n_raw = Msg.point_num; the final point is dropped as the reference does (:249). */
int32_t lvi_organize_scan(lvi_lidar *h, const lvi_livox_pt *pts, int32_t n_raw, lvi_scan_info *out);
/* the same with per-point IMU deskew: projectPointCloud's deskewPoint call (imageProjection.cpp:614) */
int32_t lvi_organize_scan_deskew(lvi_lidar *h, const lvi_livox_pt *pts, int32_t n_raw, const lvi_deskew_info *info, lvi_scan_info *out);

/* a-1..a-4  FeatureExtraction::laserCloudInfoHandler (featureExtraction.cpp:72-85):
 *      calculateSmoothness, markOccludedPoints, extractFeatures (incl. per-ring VoxelGrid). */
int32_t lvi_extract_features(lvi_lidar *h, const lvi_scan_info *in, lvi_cloud *corner, lvi_cloud *surf);

/* a-4  pcl::VoxelGrid<PointXYZI>::filter with setLeafSize(leaf,leaf,leaf)
 *      (featureExtraction.cpp:61,240-241; mapOptimization.cpp:247-250,959-964,991-997). */
int32_t lvi_voxel_downsample(lvi_lidar *h, const lvi_pt *in, int32_t n, float leaf, lvi_pt *out, int32_t out_capacity, int32_t *n_out);

/* a-4 + a-6  extractCloud's two VoxelGrid calls over the fused local map and the two
 *      KdTreeFLANN::setInputCloud calls (mapOptimization.cpp:958-965, 1322-1323).
 *      Inputs are laserCloud{Corner,Surf}FromMap (already in the map frame). */
int32_t lvi_map_set(lvi_lidar *h, const lvi_pt *corner_raw, int32_t nc, const lvi_pt *surf_raw, int32_t ns);

/* a-4(scan DS) + a-5..a-10  downsampleCurrentScan + scan2MapOptimization
 *      (mapOptimization.cpp:987-999, 1315-1375).  pose = transformTobeMapped
 *      [roll,pitch,yaw,x,y,z], in: initial guess, out: result. */
int32_t lvi_scan_to_map(lvi_lidar *h, const lvi_pt *corner, int32_t nc, const lvi_pt *surf, int32_t ns,
                        const lvi_imu_hint *imu, float pose[6], lvi_icp_result *out);

/* a-5  mapOptimization::transformPointCloud (mapOptimization.cpp:347-385) with
 *      pcl::getTransformation(x,y,z,roll,pitch,yaw); pose6 = [roll,pitch,yaw,x,y,z]. */
int32_t lvi_transform_cloud(lvi_lidar *h, const lvi_pt *in, int32_t n, const float pose6[6], lvi_pt *out);

/* ---- staged, device-resident form of the same path -----------------------
 * upload → run stages on the handle's stream → fetch.  Used by the replay
 * harness and bench so that inputs are resident in HBM when timing starts. */
int32_t lvi_scan_upload(lvi_lidar *h, const lvi_livox_pt *pts, int32_t n_raw);   /* H2D only */
/* ---- f-4: keyframe store and map assembly on the device --------------------------------------------------------------
 * saveKeyFramesAndFactor pushes the scan's DS clouds into cornerCloudKeyFrames / surfCloudKeyFrames with the optimised
 * pose (mapOptimization.cpp:1594-1599); extractCloud (:931-957) transforms every selected keyframe by its pose
 * (transformPointCloud :347-366) and concatenates them into laserCloud{Corner,Surf}FromMap.  With the clouds resident
 * on the device the per-scan upload of the raw local map (16 B x 5 M points) disappears: the node sends the ordered
 * list of selected keyframe indices (extractNearby's kd-tree radius search over the key POSES stays in the node).
 * Poses are [roll, pitch, yaw, x, y, z] as transformTobeMapped / PointTypePose. */
int32_t lvi_keyframe_add(lvi_lidar *h, const lvi_pt *corner, int32_t nc, const lvi_pt *surf, int32_t ns, const float pose[6], int32_t *index_out);
/* the same from the clouds the last lvi_scan_downsample / lvi_scan_to_map left on the device (laserCloud{Corner,Surf}LastDS) */
int32_t lvi_keyframe_add_current(lvi_lidar *h, const float pose[6], int32_t *index_out);
int32_t lvi_keyframe_set_pose(lvi_lidar *h, int32_t index, const float pose[6]);      /* correctPoses :1650-1660 */
int32_t lvi_keyframe_count(lvi_lidar *h, int32_t *n_keyframes, int32_t *n_points);
int32_t lvi_keyframes_clear(lvi_lidar *h);
/* extractCloud for the keys in the given order (duplicates allowed, as the reference's list may hold them) followed by the
 * two map VoxelGrids and the index build: equivalent to lvi_map_set(fused corner, fused surf). */
int32_t lvi_map_assemble(lvi_lidar *h, const int32_t *key_indices, int32_t n_keys);
/* The same local map — laserCloud{Corner,Surf}FromMapDS and the search index, bit for bit what lvi_map_assemble(keys)
 * returns — maintained incrementally: consecutive scans share almost all keyframes (extractNearby :894-929), so only the
 * keyframes that enter or leave the list are added to / taken from persistent per-voxel sums; nothing re-reads the millions
 * of points of the keyframes that stay.  A changed pose of a listed key (lvi_keyframe_set_pose), a duplicate in the list or
 * PCL's overflow rule falls back to the full assembly.  The raw fused clouds (LVI_DBG_MAP_*_RAW) are not produced. */
int32_t lvi_map_update(lvi_lidar *h, const int32_t *key_indices, int32_t n_keys);
int32_t lvi_scan_organize(lvi_lidar *h);                                         /* a-0 (+ f-1 when a deskew table is set) */
/* f-1: rotation table for the NEXT lvi_scan_organize calls (copied; NULL or imu_available == 0 switches deskew off). */
int32_t lvi_scan_set_deskew(lvi_lidar *h, const lvi_deskew_info *info);
int32_t lvi_scan_extract(lvi_lidar *h);                                          /* a-1..a-4 */
int32_t lvi_scan_downsample(lvi_lidar *h);                                       /* downsampleCurrentScan :987-999 */
int32_t lvi_map_upload(lvi_lidar *h, const lvi_pt *corner_raw, int32_t nc, const lvi_pt *surf_raw, int32_t ns);
int32_t lvi_map_build(lvi_lidar *h);                                             /* extractCloud DS + index build */
int32_t lvi_scan_match(lvi_lidar *h, const lvi_imu_hint *imu, float pose[6], lvi_icp_result *out);
/* [hip only] enqueue scan matching, write the 32-B pose record to device memory
 * (d_record: device pointer, e.g. a slot of the buffer RCCL all-gathers); no host sync. */
int32_t lvi_scan_match_async(lvi_lidar *h, const float pose_init[6], void *d_record);

/* [hip only] the same uploads from buffers that are already in HBM (device pointers on the
 * handle's GPU, e.g. the tensors a replay harness keeps resident); device-to-device, no host sync. */
int32_t lvi_scan_upload_device(lvi_lidar *h, const void *d_pts, int32_t n_raw);
int32_t lvi_map_upload_device(lvi_lidar *h, const void *d_corner_raw, int32_t nc, const void *d_surf_raw, int32_t ns);
/* [hip only] h reads the raw local map `owner` holds, in place: several handles of one GPU that match scans against the
 * same local map (replay harnesses: laserCloud{Corner,Surf}FromMap is one read-only cloud, mapOptimization.cpp:958-965)
 * keep ONE copy of it in HBM, which then also stays in the 256 MB memory-side cache while every handle re-voxelises it.
 * Each handle still builds its own downsampled map and index.  `owner` must hold a map and live on the same GPU; while h shares
 * it, lvi_map_upload* / lvi_map_assemble on `owner` fail with LVI_ERR_STATE and `owner` must not be destroyed (destroy the
 * sharing handles first).  A later lvi_map_upload* / lvi_map_assemble on h, or its destruction, ends the sharing (h goes back
 * to its own memory).  The oracle copies the clouds instead (same results, no such restriction). */
int32_t lvi_map_share(lvi_lidar *h, lvi_lidar *owner);

/* [hip only] the whole per-scan path in one call, for replay harnesses: scan (device pointer, Msg.point_num
 * points) → organise → features → scan DS → [re-voxelise + re-index the uploaded raw map, as the reference
 * does for every scan, when rebuild_map != 0] → scan matching from pose_init (imu_available = 0) → 32-byte
 * pose record written to d_record (device pointer).  Nothing is synchronised; the launch sequence is captured
 * once into a hipGraph per (n_raw, map size, rebuild_map) and replayed, so a call costs one graph launch. */
int32_t lvi_scan_replay_enqueue(lvi_lidar *h, const void *d_pts, int32_t n_raw, const float pose_init[6], void *d_record, int32_t rebuild_map);

/* ---- batched form [hip: one launch sequence for the whole batch; oracle: a loop] ---------------------------------
 * Up to lvi_lidar_params.batch_scans independent scans (slots 0 .. n_scans-1) go through organise → features → scan
 * DS → [per-slot re-voxelisation + re-indexing of the handle's raw local map when rebuild_map != 0, as the reference
 * does for every scan, mapOptimization.cpp:958-965, 1322-1323] → scan matching SIDE BY SIDE: every kernel of the path
 * carries the slot in blockIdx.z, so the batch costs the launches of one scan and the one-workgroup links of the
 * dependent chain (scans, the 6x6 solves) run n_scans workgroups wide.  Results are bit-identical to the single-scan
 * entry points.  The raw local map (lvi_map_upload* / lvi_map_assemble) is shared by the slots; lvi_map_build builds every
 * slot's DS map and index.  n_raw[z] = Msg.point_num of scan z (the final point is dropped, imageProjection.cpp:249). */
/* scans already in HBM, read in place (no copy): the buffers must stay valid until the batch has run */
int32_t lvi_scan_batch_bind_device(lvi_lidar *h, int32_t n_scans, const void *const *d_pts, const int32_t *n_raw);
/* scans in host memory (copied through pinned staging, no stream sync) */
int32_t lvi_scan_batch_upload(lvi_lidar *h, int32_t n_scans, const lvi_livox_pt *const *pts, const int32_t *n_raw);
/* enqueue the path for the bound / uploaded scans from pose_init[z][6] (imu_available = 0); slot z's 32-byte pose record
 * goes to d_records + 32 z (device pointer, may be NULL).  No host sync.  A device-side error of a slot travels in its
 * record's status. */
int32_t lvi_scan_batch_run(lvi_lidar *h, int32_t n_scans, const float *pose_init, void *d_records, int32_t rebuild_map);
/* wait for the batch and copy the records of slots 0 .. n_scans-1 to the host */
int32_t lvi_scan_batch_get_records(lvi_lidar *h, int32_t n_scans, lvi_pose_record *out);
/* [hip only] the slot the fetch / inspection entry points below (and the single-scan stage calls) address; default 0 */
int32_t lvi_batch_select(lvi_lidar *h, int32_t slot);

/* fetch current stage outputs (host buffers) */
int32_t lvi_get_scan_info(lvi_lidar *h, lvi_scan_info *out);
int32_t lvi_get_features(lvi_lidar *h, lvi_cloud *corner, lvi_cloud *surf);       /* cornerCloud, surfaceCloud */
int32_t lvi_get_scan_ds(lvi_lidar *h, lvi_cloud *corner_ds, lvi_cloud *surf_ds);  /* laserCloud{Corner,Surf}LastDS */
int32_t lvi_get_map_ds(lvi_lidar *h, lvi_cloud *corner_ds, lvi_cloud *surf_ds);   /* laserCloud{Corner,Surf}FromMapDS */
int32_t lvi_get_counts(lvi_lidar *h, int32_t counts[8]);
int32_t lvi_get_pose_record(lvi_lidar *h, lvi_pose_record *out);   /* waits for and returns the last scan match's record */
/* counts: [0] n extracted, [1] corners, [2] surf (after per-ring DS), [3] corner DS, [4] surf DS,
 *         [5] map corner DS, [6] map surf DS, [7] reserved */

/* ---- inspection of intermediates (parity tests) ---------------------------- */
enum {
    LVI_DBG_CURVATURE      = 1,   /* f32[n]   cloudCurvature            (featureExtraction.cpp:103) */
    LVI_DBG_PICKED_OCCL    = 2,   /* i32[n]   cloudNeighborPicked after markOccludedPoints (:113-148) */
    LVI_DBG_LABEL          = 3,   /* i32[n]   cloudLabel after extractFeatures: 1 corner, -1 surf-picked, 0 */
    LVI_DBG_PICKED_FINAL   = 4,   /* i32[n]   cloudNeighborPicked after extractFeatures */
    LVI_DBG_CORNER_INDEX   = 5,   /* i32[C]   index into extractedCloud of every corner, in output order */
    LVI_DBG_VOXEL_KEYS     = 6,   /* i32[P]   per-input-point voxel idx of the last lvi_voxel_downsample call */
    LVI_DBG_VOXEL_CELLS    = 7,   /* i32[V]   distinct idx, ascending = output order, same call */
    LVI_DBG_VOXEL_COUNTS   = 8,   /* i32[V]   points per output voxel, same call */
    LVI_DBG_ICP_JTJ        = 9,   /* f32[iters*27] 21 upper-triangular AtA + 6 AtB per iteration */
    LVI_DBG_ICP_POSE_TRACE = 10,  /* f32[(iters+1)*6] transformTobeMapped before iteration k (and after the last) */
    LVI_DBG_ICP_CYCLES     = 12,  /* i64[16] [hip only] ([8..12]: solve kernel: partial sums, combine, solve, pose, total) shader cycles of workgroup 0 of the last residual launch: pose load, KNN scan,
                                     top-5 merge, residual math, row reduction, total; [6] = candidates scanned by lane 0 */
    LVI_DBG_MAP_CORNER_RAW = 13,  /* lvi_pt[] laserCloudCornerFromMap as uploaded / assembled */
    LVI_DBG_MAP_SURF_RAW   = 14,  /* lvi_pt[] laserCloudSurfFromMap */
    LVI_DBG_FEAT_CYCLES    = 11   /* i64[8] [hip only] shader cycles of ring 0's sector kernel by phase: load, compact, rank, walk,
                                     fixed-point set-up, fixed-point rounds, number of rounds, apply+store */
};
int32_t lvi_debug_get(lvi_lidar *h, int32_t what, void *dst, int64_t capacity_bytes, int64_t *n_bytes);

/* a-6  5-NN of arbitrary query points against the current DS map (which: 0 corner, 1 surf).
 *      idx: [nq*5] indices into laserCloud*FromMapDS; sqd: [nq*5] squared distances, ascending.
 *      For a query whose 5th neighbour is not closer than 1 m (rejected at
 *      mapOptimization.cpp:1025,1121) idx is -1 and sqd is +inf from the first slot that is not < 1.0. */
int32_t lvi_debug_knn(lvi_lidar *h, int32_t which, const lvi_pt *queries, int32_t nq, int32_t *idx, float *sqd);

/* a-7/a-8  one pass of cornerOptimization / surfOptimization at a given pose over the
 *      current DS scan: coeff[nq] (coeffSel entry), flag[nq] (laserCloudOri*Flag). */
int32_t lvi_debug_residuals(lvi_lidar *h, int32_t which, const float pose[6], lvi_pt *coeff, uint8_t *flag, int32_t capacity, int32_t *n);

/* ---- kernel timing (HIP events on the handle's stream) ---------------------
 * [hip only]  enable, run stages, then read accumulated per-kernel time. */
typedef struct lvi_kernel_stat {
    char     name[48];
    int64_t  launches;
    double   total_ms;
    double   bytes_alg;          /* algorithmic bytes summed over launches (DESIGN.md per-kernel formulas) */
} lvi_kernel_stat;
int32_t lvi_prof_enable(lvi_lidar *h, int32_t on);
int32_t lvi_prof_reset(lvi_lidar *h);
int32_t lvi_prof_read(lvi_lidar *h, lvi_kernel_stat *stats, int32_t capacity, int32_t *n);

/* =========================================================================== */
/* feature_tracker                                                             */
/* =========================================================================== */

/* parameters.cpp:53-110 globals read on the path + cv:: defaults fixed by the call
 * sites feature_tracker.cpp:113 and :166. */
typedef struct lvi_tracker_params {
    int32_t max_width, max_height;   /* capacity; yaml 1024x576, benchmark 1280x720 */
    int32_t max_cnt;                 /* MAX_CNT   yaml 150 */
    double  min_dist;                /* MIN_DIST  yaml 20 (int in the reference, passed as double minDistance) */
    int32_t lk_win;                  /* 21  (cv::Size(21,21)) */
    int32_t lk_max_level;            /* 3   → 4 pyramid levels */
    int32_t lk_max_iters;            /* 30  (TermCriteria default) */
    double  lk_eps;                  /* 0.01 (TermCriteria::epsilon, double) */
    float   lk_min_eig_threshold;    /* 1e-4 */
    double  gftt_quality;            /* 0.01 (double qualityLevel) */
    int32_t max_features;            /* capacity of the per-frame point arrays */
} lvi_tracker_params;

/* f-3: MEI (unified omnidirectional) camera, camera_model CataCamera::Parameters (CataCamera.cc:14-40). */
typedef struct lvi_mei_params {
    double xi, k1, k2, p1, p2, gamma1, gamma2, u0, v0;
} lvi_mei_params;

typedef struct lvi_tracker lvi_tracker;    /* opaque */

void    lvi_tracker_params_default(lvi_tracker_params *p);
int32_t lvi_tracker_create(const lvi_tracker_params *p, int32_t device, lvi_tracker **out);
void    lvi_tracker_destroy(lvi_tracker *t);
int32_t lvi_tracker_sync(lvi_tracker *t);

/* a-11  cv::calcOpticalFlowPyrLK(cur_img, forw_img, cur_pts, forw_pts, status, err,
 *       cv::Size(21,21), 3)  (feature_tracker.cpp:113). */
int32_t lvi_lk_track(lvi_tracker *t, const uint8_t *prev, const uint8_t *next, int32_t w, int32_t h, int32_t stride,
                     const float *prev_xy, int32_t n, float *next_xy, uint8_t *status, float *err);

/* a-12  cv::goodFeaturesToTrack(forw_img, n_pts, max_corners, quality, min_dist, mask)
 *       (feature_tracker.cpp:166).  mask may be NULL (all 255). */
int32_t lvi_good_features(lvi_tracker *t, const uint8_t *img, const uint8_t *mask, int32_t w, int32_t h, int32_t stride,
                          int32_t max_corners, double quality, double min_dist, float *xy, int32_t xy_capacity, int32_t *n_out);

/* staged form mirroring FeatureTracker::readImage's image rotation
 * (feature_tracker.cpp:94-101, 200-204): push makes the new image "forw"
 * (its pyramid is built once and kept resident), the previous forw becomes "cur". */
/* f-2  cv::createCLAHE(3.0, cv::Size(8, 8))->apply(_img, img)  (feature_tracker.cpp:86-90, EQUALIZE = 1).
 *      One-call form: host image in, host image out. */
int32_t lvi_clahe(lvi_tracker *t, const uint8_t *img, int32_t w, int32_t h, int32_t stride,
                  double clip_limit, int32_t tiles_x, int32_t tiles_y, uint8_t *out, int32_t out_stride);
/* f-2 staged: every later lvi_tracker_push_image equalises the frame on the device before the pyramid is built
 *      (readImage's `if (EQUALIZE)` branch); on = 0 restores `img = _img`. */
int32_t lvi_tracker_set_equalize(lvi_tracker *t, int32_t on, double clip_limit, int32_t tiles_x, int32_t tiles_y);

/* f-3  undistortedPoints(): m_camera->liftProjective(a, b); cur_un_pts = (b.x/b.z, b.y/b.z)
 *      (feature_tracker.cpp:298-311; CataCamera::liftProjective CataCamera.cc:556-626, distortion :766-783).
 *      xy / un_xy: n packed (x, y) f32 pairs, as std::vector<cv::Point2f>. */
int32_t lvi_undistort_points(lvi_tracker *t, const lvi_mei_params *cam, const float *xy, int32_t n, float *un_xy);

int32_t lvi_tracker_push_image(lvi_tracker *t, const uint8_t *img, int32_t w, int32_t h, int32_t stride);
int32_t lvi_tracker_set_points(lvi_tracker *t, const float *cur_xy, int32_t n);                   /* H2D cur_pts */
int32_t lvi_tracker_run_lk(lvi_tracker *t);                                                        /* cur → forw, device-resident */
int32_t lvi_tracker_get_lk(lvi_tracker *t, float *forw_xy, uint8_t *status, float *err, int32_t capacity, int32_t *n);
int32_t lvi_tracker_set_mask(lvi_tracker *t, const uint8_t *mask, int32_t w, int32_t h, int32_t stride); /* NULL → all 255 */
int32_t lvi_tracker_run_gftt(lvi_tracker *t, int32_t max_corners);                                 /* on forw */
int32_t lvi_tracker_get_gftt(lvi_tracker *t, float *xy, int32_t capacity, int32_t *n);
/* a-13 setMask (feature_tracker.cpp:36-69), the raster half: mask = 255, then cv::circle(mask, pt, radius, 0, -1) around every listed
 *      point (centres rounded as Mat::at(Point2f) / cv::circle round them).  The caller keeps the ordering by track_cnt and the
 *      "is this point still free" walk (:50-67) — 150 points, host — and hands over the points it kept: 1.2 KB cross the bus instead
 *      of the W x H mask.  Enqueued, not synchronised. */
int32_t lvi_tracker_set_mask_circles(lvi_tracker *t, const float *centers_xy, int32_t n, int32_t radius);
/* a-12 cv::goodFeaturesToTrack(forw_img, n_pts, MAX_CNT - forw_pts.size(), 0.01, MIN_DIST, mask) (feature_tracker.cpp:166), enqueued
 *      only: lvi_tracker_finish_frame fetches the corners. */
int32_t lvi_tracker_run_gftt_async(lvi_tracker *t, int32_t max_corners);
/* The end of readImage (feature_tracker.cpp:166-205) in ONE read of the device: the corners of a pending lvi_tracker_run_gftt_async
 * (n_new, new_xy; none pending: n_new = 0) and — cam != NULL — undistortedPoints() (f-3, :298-311) of cur_pts of the next frame
 * = [kept_xy ; new corners] (addPoints :71-79 appends them in this order): un_xy receives n_kept + n_new (x, y) pairs. */
int32_t lvi_tracker_finish_frame(lvi_tracker *t, const lvi_mei_params *cam, const float *kept_xy, int32_t n_kept,
                                 float *new_xy, int32_t new_capacity, int32_t *n_new, float *un_xy);

enum {
    LVI_TDBG_PYRAMID_L1 = 1,   /* u8 level-1 image of forw (pyrDown) */
    LVI_TDBG_PYRAMID_L2 = 2,
    LVI_TDBG_PYRAMID_L3 = 3,
    LVI_TDBG_MINEIG     = 4,   /* f32[h*w] cornerMinEigenVal map of forw */
    LVI_TDBG_GFTT_NCAND = 5,   /* i32[1]  number of local-maximum candidates before the distance filter */
    LVI_TDBG_MASK       = 6    /* u8[h*w] the mask goodFeaturesToTrack reads (lvi_tracker_set_mask / lvi_tracker_set_mask_circles) */
};
int32_t lvi_tracker_debug_get(lvi_tracker *t, int32_t what, void *dst, int64_t capacity_bytes, int64_t *n_bytes);

int32_t lvi_tracker_prof_enable(lvi_tracker *t, int32_t on);
int32_t lvi_tracker_prof_reset(lvi_tracker *t);
int32_t lvi_tracker_prof_read(lvi_tracker *t, lvi_kernel_stat *stats, int32_t capacity, int32_t *n);

#ifdef __cplusplus
}
#endif
#endif /* LVI_HOTPATH_H */
