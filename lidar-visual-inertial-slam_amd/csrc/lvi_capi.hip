#include <atomic>
// C-ABI of the lidar half (include/lvi_hotpath.h) over the HIP stages.  Host-side only:
// argument checks, H2D/D2H, stage ordering, error mapping.  No CPU fallback exists: every
// compute entry point enqueues HIP kernels or fails.
#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>

#include "lvi_lidar.hpp"

namespace lvi {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }

// ---- Profiler ---------------------------------------------------------------------------------
int Profiler::name_id(const char* n)
{
    for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return (int)i;
    names.emplace_back(n); launches.push_back(0); total_ms.push_back(0.0); bytes.push_back(0.0);
    return (int)names.size() - 1;
}
hipEvent_t Profiler::get_event()
{
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void Profiler::collect()
{
    for (auto& r : recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { launches[r.name_id]++; total_ms[r.name_id] += ms; bytes[r.name_id] += r.bytes; }
        pool.push_back(r.a); pool.push_back(r.b);
    }
    recs.clear();
}
void Profiler::reset()
{
    collect();
    std::fill(launches.begin(), launches.end(), 0); std::fill(total_ms.begin(), total_ms.end(), 0.0); std::fill(bytes.begin(), bytes.end(), 0.0);
}
Profiler::~Profiler()
{
    for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : pool) (void)hipEventDestroy(e);
}

// defined in lvi_voxel.hip
void voxel_debug_fetch(const Ctx& ctx, const VoxelPlan& p, int n_in, std::vector<int32_t>& keys, std::vector<int32_t>& cells, std::vector<int32_t>& counts);

}  // namespace lvi

using namespace lvi;

struct lvi_lidar {
    LidarDev d;                                     // slot 0: owns the streams, the profiler, the keyframe store and the raw map
    std::vector<std::unique_ptr<LidarDev>> more;    // batch slots 1 .. batch_scans-1 (lvi_scan_batch_*)
    std::vector<LidarDev*> slots;                   // [batch_scans] = &d, more[0], …
    int sel = 0;                                    // slot the fetch / inspection entry points read (lvi_batch_select)
    LidarDev& cur() { return *slots[sel]; }
    std::vector<int32_t> vkeys, vcells, vcounts;    // debug of the last lvi_voxel_downsample
    bool vdbg_pending = false; int vdbg_n = 0;      // … not fetched yet (lvi_debug_get does it)
    // lvi_voxel_downsample of a small cloud (<= VOX_TINY points: the node's key-pose grid): pinned staging of the one-launch form
    lvi_pt *vt_in = nullptr, *vt_out = nullptr; int *vt_hdr = nullptr, *vt_cells = nullptr, *vt_counts = nullptr, *vt_keys = nullptr;
    bool have_icp_host = false;
    bool icp_host_full = false;            // the IcpState mirror (traces for the debug views) has been fetched for the last match
    lvi_lidar* share_owner = nullptr;               // lvi_map_share: the handle whose raw map this one reads
    std::atomic<int> shared_by{0};                  // handles that read this one's raw map (they may live on other host threads): it must not change while > 0
    std::vector<lvi_lidar*> sharers;                // … who they are (g_share_mu): an owner that is destroyed first sends them back to their own memory
};

namespace {

std::mutex g_share_mu;                              // the sharer lists of lvi_map_share (handles may live on different host threads)

int32_t fail(int32_t code, const std::string& msg) { set_error(msg); return code; }

template <class F>
int32_t guarded(lvi_lidar* h, F&& f)
{
    try {
        if (h) LVI_HIP(hipSetDevice(h->d.device));
        return f();
    } catch (const HipError& e) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", e.what, hipGetErrorString(e.e), e.file, e.line);
        return fail(LVI_ERR_HIP, buf);
    } catch (const std::exception& e) {
        return fail(LVI_ERR_HIP, e.what());
    }
}

void sync(LidarDev& d)
{
    if (d.map_pending) join_map(d);
    LVI_HIP(hipStreamSynchronize(d.ctx.stream));
}

template <class T>
void d2h(LidarDev& d, T* dst, const T* src, size_t n)
{
    join_map(d);       // a pending map build (second stream) must be visible to anything the host reads or rewrites
    if (n) LVI_HIP(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, d.ctx.stream));
}
template <class T>
void h2d(LidarDev& d, T* dst, const T* src, size_t n)
{
    join_map(d);
    if (n) LVI_HIP(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, d.ctx.stream));
}

int read_int(LidarDev& d, const int* p)
{
    int v = 0;
    d2h(d, &v, p, 1);
    sync(d);
    return v;
}

int32_t dev_status_code(int st)
{
    if (st & DEV_ERR_SECTOR_TOO_LARGE) return fail(LVI_ERR_CAPACITY, "a ring sector exceeds FEAT_SEG_CAP points (Horizon_SCAN too large for the LDS-resident sector kernel)");
    if (st & DEV_ERR_SECTOR_HANDOVER) return fail(LVI_ERR_HIP, "sector kernel: a workgroup never received its predecessor's hand-over word");
    if (st & DEV_ERR_GRID_TOO_LARGE) return fail(LVI_ERR_CAPACITY, "local map extent too large for the KNN grid");
    return LVI_OK;
}
int32_t check_dev_status(LidarDev& d)
{
    int w[2] = {0, 0};
    d2h(d, w, d.d_status, 2);
    sync(d);
    const int st = w[0] | w[1];
    if (st & DEV_ERR_SECTOR_TOO_LARGE) return fail(LVI_ERR_CAPACITY, "a ring sector exceeds FEAT_SEG_CAP points (Horizon_SCAN too large for the LDS-resident sector kernel)");
    if (st & DEV_ERR_SECTOR_HANDOVER) return fail(LVI_ERR_HIP, "sector kernel: a workgroup never received its predecessor's hand-over word");
    if (st & DEV_ERR_GRID_TOO_LARGE) return fail(LVI_ERR_CAPACITY, "local map extent too large for the KNN grid");
    return LVI_OK;
}

int32_t fetch_cloud(LidarDev& d, const lvi_pt* src, int n, lvi_cloud* dst)
{
    if (!dst) return LVI_OK;
    dst->n = n;
    if (dst->capacity < n || (!dst->pts && n > 0)) return fail(LVI_ERR_CAPACITY, "cloud capacity too small");
    d2h(d, dst->pts, src, (size_t)n);
    sync(d);
    return LVI_OK;
}

struct Counts { int n, ncorner, nsurf, ncds, nsds, mcds, msds; };
Counts read_counts(LidarDev& d)
{
    Counts c{};
    int a = 0, b = 0, ring[MAX_N_SCAN + 1] = {0}, sc[3] = {0}, mp[3] = {0};
    if (d.have_org) d2h(d, &a, d.d_n, 1);
    if (d.have_feat) { d2h(d, &b, d.d_ncorner, 1); d2h(d, ring, d.voxRing.d_nout, (size_t)d.P.N_SCAN + 1); }
    if (d.have_ds) d2h(d, sc, d.voxScan.d_nout, 3);
    if (d.have_map) d2h(d, mp, d.voxMap.d_nout, 3);
    sync(d);
    c.n = a; c.ncorner = b; c.nsurf = ring[d.P.N_SCAN]; c.ncds = sc[0]; c.nsds = sc[1]; c.mcds = mp[0]; c.msds = mp[1];
    return c;
}

template <class T, class U>
int32_t dbg_out(const std::vector<T>& v, U, void* dst, int64_t cap, int64_t* n_bytes)
{
    const int64_t bytes = (int64_t)(v.size() * sizeof(T));
    if (n_bytes) *n_bytes = bytes;
    if (!dst) return LVI_OK;
    if (cap < bytes) return fail(LVI_ERR_CAPACITY, "debug buffer too small");
    if (bytes) memcpy(dst, v.data(), (size_t)bytes);
    return LVI_OK;
}

}  // namespace

extern "C" {

int32_t lvi_abi_version(void) { return LVI_ABI_VERSION; }
const char* lvi_backend(void) { return "hip-gfx950"; }
const char* lvi_last_error(void) { return lvi::g_err.c_str(); }

void lvi_lidar_params_default(lvi_lidar_params* p)
{
    memset(p, 0, sizeof(*p));
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

int32_t lvi_lidar_create(const lvi_lidar_params* p, int32_t device, lvi_lidar** out)
{
    if (!p || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (p->N_SCAN <= 0 || p->N_SCAN > MAX_N_SCAN || p->Horizon_SCAN <= 0 || p->downsampleRate <= 0)
        return fail(LVI_ERR_INVALID_ARG, "bad scan geometry (N_SCAN must be 1..32)");
    if (p->icp_max_iters < 0) return fail(LVI_ERR_INVALID_ARG, "icp_max_iters < 0");
    if (p->max_map_points > (1 << 25) || p->max_raw_points > (1 << 25)) return fail(LVI_ERR_INVALID_ARG, "capacities above 2^25 points are not supported");
    if (p->voxel_mode < 0 || p->voxel_mode > 2) return fail(LVI_ERR_INVALID_ARG, "voxel_mode must be 0 (auto), 1 (sorted) or 2 (binned)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(LVI_ERR_NO_DEVICE, "no HIP device: the HIP path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(LVI_ERR_NO_DEVICE, "device index out of range");
    if (p->batch_scans < 0 || p->batch_scans > LVI_MAX_BATCH) return fail(LVI_ERR_INVALID_ARG, "batch_scans must be 0..LVI_MAX_BATCH");
    lvi_lidar* h = new lvi_lidar();
    h->slots.push_back(&h->d);
    h->d.P = *p; h->d.device = device;
    const int S = std::max(p->batch_scans, 1);
    if (S > 1 && !getenv("LVI_BATCH_TWO_STREAMS")) h->d.P.map_on_main_stream = 1;      // a batch fills the chip by itself; one stream keeps every slot's work in one order
    int32_t st = guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        LVI_HIP(hipStreamCreateWithFlags(&d.ctx.stream, hipStreamNonBlocking));
        LVI_HIP(hipStreamCreateWithFlags(&d.ctx2.stream, hipStreamNonBlocking));
        LVI_HIP(hipEventCreateWithFlags(&d.evMain, hipEventDisableTiming));
        LVI_HIP(hipEventCreateWithFlags(&d.evMap, hipEventDisableTiming));
        d.ctx.prof = &d.prof; d.ctx2.prof = &d.prof;
        lidar_allocate(d);
        for (int z = 1; z < S; z++) {
            h->more.emplace_back(new LidarDev());
            LidarDev& q = *h->more.back();
            q.P = d.P; q.device = device; q.ctx = d.ctx; q.ctx2 = d.ctx2; q.map_owner = &d;
            h->slots.push_back(&q);
            lidar_allocate(q);
        }
        return LVI_OK;
    });
    if (st != LVI_OK) { lvi_lidar_destroy(h); return st; }
    *out = h;
    return LVI_OK;
}

static void release_slot(LidarDev& d)
{
    d.voxRing.release(); d.voxScan.release(); d.voxMap.release(); d.voxGen.release();
    d.arena.release();
    if (d.h_icp) (void)hipHostFree(d.h_icp);
    if (d.h_gn_feat) (void)hipHostFree(d.h_gn_feat);
    if (d.h_res) (void)hipHostFree(d.h_res);
    if (d.h_kfSeg) (void)hipHostFree(d.h_kfSeg);
    if (d.inc.h_pieces) (void)hipHostFree(d.inc.h_pieces);
    if (d.inc.h_status) (void)hipHostFree(d.inc.h_status);
    if (d.inc.h_active) (void)hipHostFree(d.inc.h_active);
    for (int s = 0; s < 2; s++) {
        if (d.h_raw[s]) (void)hipHostFree(d.h_raw[s]);
        if (d.ev_raw[s]) (void)hipEventDestroy(d.ev_raw[s]);
    }
    if (d.graphExec) (void)hipGraphExecDestroy(d.graphExec);
}

static void leave_owner(lvi_lidar* h);
static void unshare_map(lvi_lidar* h);

void lvi_lidar_destroy(lvi_lidar* h)
{
    if (!h) return;
    LidarDev& d = h->d;
    (void)hipSetDevice(d.device);
    leave_owner(h);
    // an owner that goes first: its sharers return to their own (empty) raw-map memory before this handle's is freed
    for (;;) {
        lvi_lidar* q = nullptr;
        { std::lock_guard<std::mutex> lk(g_share_mu); if (!h->sharers.empty()) q = h->sharers.back(); }
        if (!q) break;
        try { (void)hipSetDevice(q->d.device); unshare_map(q); } catch (...) { leave_owner(q); }
    }
    (void)hipSetDevice(d.device);
    if (d.ctx.stream) { (void)hipStreamSynchronize(d.ctx.stream); }
    if (d.ctx2.stream) { (void)hipStreamSynchronize(d.ctx2.stream); }
    d.prof.collect();
    for (auto& q : h->more) release_slot(*q);
    release_slot(d);
    if (h->vt_in) { (void)hipHostFree(h->vt_in); (void)hipHostFree(h->vt_out); (void)hipHostFree(h->vt_hdr); (void)hipHostFree(h->vt_cells); (void)hipHostFree(h->vt_counts); (void)hipHostFree(h->vt_keys); }
    if (d.ctx.stream) (void)hipStreamDestroy(d.ctx.stream);
    if (d.ctx2.stream) (void)hipStreamDestroy(d.ctx2.stream);
    for (int s = 0; s < LVI_LIDAR_MARKS; s++) if (d.evMark[s]) (void)hipEventDestroy(d.evMark[s]);
    if (d.evMain) (void)hipEventDestroy(d.evMain);
    if (d.evMap) (void)hipEventDestroy(d.evMap);
    if (d.evMapDeps) (void)hipEventDestroy(d.evMapDeps);
    delete h;
}

int32_t lvi_lidar_sync(lvi_lidar* h)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    return guarded(h, [&]() -> int32_t { sync(h->cur()); return LVI_OK; });
}

int32_t lvi_lidar_mark(lvi_lidar* h, int32_t slot)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    if (slot < 0 || slot >= LVI_LIDAR_MARKS) return fail(LVI_ERR_INVALID_ARG, "mark slot out of range");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        if (!d.evMark[slot]) LVI_HIP(hipEventCreateWithFlags(&d.evMark[slot], hipEventDisableTiming));
        join_map(d);                                       // a map build still on its own stream is part of "everything enqueued so far"
        LVI_HIP(hipEventRecord(d.evMark[slot], d.ctx.stream));
        return LVI_OK;
    });
}

int32_t lvi_lidar_wait_mark(lvi_lidar* h, int32_t slot)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    if (slot < 0 || slot >= LVI_LIDAR_MARKS) return fail(LVI_ERR_INVALID_ARG, "mark slot out of range");
    return guarded(h, [&]() -> int32_t {
        if (h->d.evMark[slot]) LVI_HIP(hipEventSynchronize(h->d.evMark[slot]));
        return LVI_OK;
    });
}

// ---- staged form ------------------------------------------------------------------------------
int32_t lvi_scan_upload(lvi_lidar* h, const lvi_livox_pt* pts, int32_t n_raw)
{
    if (!h || (n_raw > 0 && !pts)) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (n_raw > h->cur().raw_cap) return fail(LVI_ERR_CAPACITY, "n_raw exceeds max_raw_points");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        d.raw_bound = nullptr;                      // (a buffer bound by lvi_scan_batch_bind_device is no longer this slot's scan)
        d.n_raw = n_raw > 0 ? n_raw - 1 : 0;        // moveFromCustomMsg: i < point_num-1 (imageProjection.cpp:249)
        // the caller's (pageable) message is copied to pinned staging on the host and uploaded from there: the call
        // returns without a stream sync, and the runtime never has to pin / stage the user's memory itself
        const int slot = (d.raw_slot ^= 1);
        LVI_HIP(hipEventSynchronize(d.ev_raw[slot]));
        std::memcpy(d.h_raw[slot], pts, sizeof(lvi_livox_pt) * (size_t)d.n_raw);
        h2d(d, d.raw, d.h_raw[slot], (size_t)d.n_raw);
        LVI_HIP(hipEventRecord(d.ev_raw[slot], d.ctx.stream));
        LVI_HIP(hipMemsetAsync(d.d_status, 0, sizeof(int), d.ctx.stream));
        d.have_raw = true; d.have_org = d.have_feat = d.have_ds = false;
        return LVI_OK;
    });
}
int32_t lvi_scan_upload_device(lvi_lidar* h, const void* d_pts, int32_t n_raw)
{
    if (!h || (n_raw > 0 && !d_pts)) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (n_raw > h->cur().raw_cap) return fail(LVI_ERR_CAPACITY, "n_raw exceeds max_raw_points");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        d.raw_bound = nullptr;
        d.n_raw = n_raw > 0 ? n_raw - 1 : 0;
        if (d.n_raw) LVI_HIP(hipMemcpyAsync(d.raw, d_pts, sizeof(lvi_livox_pt) * (size_t)d.n_raw, hipMemcpyDeviceToDevice, d.ctx.stream));
        LVI_HIP(hipMemsetAsync(d.d_status, 0, sizeof(int), d.ctx.stream));
        d.have_raw = true; d.have_org = d.have_feat = d.have_ds = false;
        return LVI_OK;
    });
}
// A captured launch sequence (lvi_scan_replay_enqueue) froze the raw map's pointers, sizes and — with map_plan_cache — whether the
// plan passes are part of it: whatever rewrites the raw map drops it; the next lvi_scan_replay_enqueue captures again.
static void drop_graph(lvi_lidar* h)
{
    if (h->d.graphExec) { (void)hipGraphExecDestroy(h->d.graphExec); h->d.graphExec = nullptr; }
}
// every slot of h reads the raw local map at (c, s): the handle's own memory, or another handle's (lvi_map_share)
static void bind_raw_map(lvi_lidar* h, lvi_pt* c, lvi_pt* s)
{
    drop_graph(h);
    for (LidarDev* q : h->slots) {
        q->mapCornerRaw = c; q->mapSurfRaw = s;
        const VoxSegStatic st[2] = {VoxSegStatic{c, nullptr, q->mapCornerDS, q->P.mappingCornerLeafSize}, VoxSegStatic{s, nullptr, q->mapSurfDS, q->P.mappingSurfLeafSize}};
        q->voxMap.set_static(q->ctx, st);
    }
}
static void leave_owner(lvi_lidar* h)
{
    std::lock_guard<std::mutex> lk(g_share_mu);
    if (!h->share_owner) return;
    auto& v = h->share_owner->sharers;
    v.erase(std::remove(v.begin(), v.end(), h), v.end());
    h->share_owner->shared_by--;
    h->share_owner = nullptr;
}
// a handle that shares another one's map goes back to its own memory before anything writes a raw map through it
static void unshare_map(lvi_lidar* h)
{
    if (h->d.mapCornerRaw == h->d.mapCornerOwn) return;
    leave_owner(h);
    join_map(h->d); sync(h->d);
    bind_raw_map(h, h->d.mapCornerOwn, h->d.mapSurfOwn);
    h->d.have_map_raw = false; h->d.n_map_corner = h->d.n_map_surf = 0; h->d.voxMap.bbox_cached = false;
    for (LidarDev* q : h->slots) q->have_map = false;
}

int32_t lvi_map_share(lvi_lidar* h, lvi_lidar* owner)
{
    if (!h || !owner || h == owner) return fail(LVI_ERR_INVALID_ARG, "bad handles");
    if (!owner->d.have_map_raw) return fail(LVI_ERR_STATE, "the owner holds no map");
    if (owner->d.device != h->d.device) return fail(LVI_ERR_INVALID_ARG, "handles on different GPUs");
    if (owner->d.mapCornerRaw != owner->d.mapCornerOwn) return fail(LVI_ERR_STATE, "the owner itself shares a map");
    if (h->shared_by > 0) return fail(LVI_ERR_STATE, "other handles read this handle's map: it cannot share another one's");
    if (owner->d.n_map_corner > h->d.map_cap || owner->d.n_map_surf > h->d.map_cap) return fail(LVI_ERR_CAPACITY, "map exceeds max_map_points");
    return guarded(h, [&]() -> int32_t {
        join_map(owner->d); sync(owner->d);                         // the owner's upload / assembly has landed
        join_map(h->d); sync(h->d);
        leave_owner(h);
        { std::lock_guard<std::mutex> lk(g_share_mu); h->share_owner = owner; owner->shared_by++; owner->sharers.push_back(h); }
        bind_raw_map(h, owner->d.mapCornerRaw, owner->d.mapSurfRaw);
        LidarDev& d = h->d;
        d.n_map_corner = owner->d.n_map_corner; d.n_map_surf = owner->d.n_map_surf; d.have_map_raw = true; d.voxMap.bbox_cached = false;
        d.inc_ready = false;
        for (LidarDev* q : h->slots) q->have_map = false;
        return LVI_OK;
    });
}

int32_t lvi_map_upload_device(lvi_lidar* h, const void* c, int32_t nc, const void* s, int32_t ns)
{
    if (!h || nc < 0 || ns < 0 || (nc > 0 && !c) || (ns > 0 && !s)) return fail(LVI_ERR_INVALID_ARG, "bad map arguments");
    if (nc > h->d.map_cap || ns > h->d.map_cap) return fail(LVI_ERR_CAPACITY, "map exceeds max_map_points");
    if (h->shared_by > 0) return fail(LVI_ERR_STATE, "other handles read this handle's map (lvi_map_share)");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        unshare_map(h);
        drop_graph(h);
        join_map(d);
        if (nc) LVI_HIP(hipMemcpyAsync(d.mapCornerRaw, c, sizeof(lvi_pt) * (size_t)nc, hipMemcpyDeviceToDevice, d.ctx.stream));
        if (ns) LVI_HIP(hipMemcpyAsync(d.mapSurfRaw, s, sizeof(lvi_pt) * (size_t)ns, hipMemcpyDeviceToDevice, d.ctx.stream));
        d.n_map_corner = nc; d.n_map_surf = ns; d.have_map_raw = true; d.voxMap.bbox_cached = false; for (LidarDev* q : h->slots) q->have_map = false;
        return LVI_OK;
    });
}
int32_t lvi_scan_organize(lvi_lidar* h)
{
    if (!h || !h->cur().have_raw) return fail(LVI_ERR_STATE, "no scan uploaded");
    return guarded(h, [&]() -> int32_t { stage_organize(h->cur()); h->cur().have_org = true; h->cur().have_feat = h->cur().have_ds = false; return LVI_OK; });
}
int32_t lvi_scan_extract(lvi_lidar* h)
{
    if (!h || !h->cur().have_org) return fail(LVI_ERR_STATE, "scan not organised");
    return guarded(h, [&]() -> int32_t { stage_extract(h->cur()); h->cur().have_feat = true; h->cur().have_ds = false; return LVI_OK; });
}
int32_t lvi_scan_downsample(lvi_lidar* h)
{
    if (!h || !h->cur().have_feat) return fail(LVI_ERR_STATE, "features not extracted");
    return guarded(h, [&]() -> int32_t { stage_downsample(h->cur()); h->cur().have_ds = true; return LVI_OK; });
}
int32_t lvi_map_upload(lvi_lidar* h, const lvi_pt* c, int32_t nc, const lvi_pt* s, int32_t ns)
{
    if (!h || nc < 0 || ns < 0 || (nc > 0 && !c) || (ns > 0 && !s)) return fail(LVI_ERR_INVALID_ARG, "bad map arguments");
    if (nc > h->d.map_cap || ns > h->d.map_cap) return fail(LVI_ERR_CAPACITY, "map exceeds max_map_points");
    if (h->shared_by > 0) return fail(LVI_ERR_STATE, "other handles read this handle's map (lvi_map_share)");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        unshare_map(h);
        drop_graph(h);
        h2d(d, d.mapCornerRaw, c, (size_t)nc); h2d(d, d.mapSurfRaw, s, (size_t)ns);
        sync(d);
        d.n_map_corner = nc; d.n_map_surf = ns; d.have_map_raw = true; d.voxMap.bbox_cached = false; for (LidarDev* q : h->slots) q->have_map = false;
        return LVI_OK;
    });
}
int32_t lvi_map_build(lvi_lidar* h)
{
    if (!h || !h->d.have_map_raw) return fail(LVI_ERR_STATE, "no map uploaded");
    // a batch handle re-voxelises and re-indexes the (shared) raw map for every slot, in one launch sequence
    return guarded(h, [&]() -> int32_t { stage_map_build(Slots{h->slots.data(), (int)h->slots.size()}); return LVI_OK; });
}

int32_t lvi_scan_match_async(lvi_lidar* h, const float pose_init[6], void* d_record)
{
    if (!h || !pose_init) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->cur().have_ds) return fail(LVI_ERR_STATE, "scan not downsampled");
    return guarded(h, [&]() -> int32_t { set_pose_init(h->cur(), pose_init); stage_scan_match_enqueue(h->cur(), nullptr, d_record); h->have_icp_host = false; return LVI_OK; });
}

int32_t lvi_scan_match(lvi_lidar* h, const lvi_imu_hint* imu, float pose[6], lvi_icp_result* out)
{
    if (!h || !pose || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->cur().have_ds) return fail(LVI_ERR_STATE, "scan not downsampled");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        set_pose_init(d, pose);
        int nq[3] = {0, 0, 0}, dw[2] = {0, 0};
        // With the reference's break rule (:1325-1337) the loop usually ends after 3 - 6 of its <= 20 iterations, and a launch that
        // finds the loop over still costs its dispatch: the iterations go out in chunks of six, each followed by the finish step,
        // and the host looks at the state before it sends the next chunk (the finish step leaves the loop state alone).
        const int max_it = std::min(d.P.icp_max_iters, LVI_ICP_MAX_ITERS);
        const int chunk = d.P.icp_disable_break ? std::max(max_it, 1) : 6;
        for (int it0 = 0;; it0 += chunk) {
            const int it1 = std::min(it0 + chunk, max_it);
            stage_scan_match_enqueue(d, imu, nullptr, it0, it1);
            sync(d);                                                  // the finish step wrote the result block (counts and device status words
            const IcpHostResult& r = *d.h_res;                        // ride along) into pinned host memory: one wait per chunk, no copies
            for (int k = 0; k < 3; k++) nq[k] = r.nq[k];
            dw[0] = r.dw[0]; dw[1] = r.dw[1];
            if (it1 >= max_it || r.done || r.status != LVI_OK || (dw[0] | dw[1])) break;
        }
        h->have_icp_host = true; h->icp_host_full = false;
        int32_t st = dev_status_code(dw[0] | dw[1]); if (st) return st;
        const IcpHostResult& s = *d.h_res;
        memset(out, 0, sizeof(*out));
        out->status = s.final_status; out->iters = s.iters; out->converged = s.converged;
        out->degenerate = s.degenerate; out->n_corner_ds = nq[0]; out->n_surf_ds = nq[1];
        for (int i = 0; i < LVI_ICP_MAX_ITERS; i++) out->n_sel[i] = s.n_sel[i];
        for (int k = 0; k < 6; k++) { out->pose[k] = s.final_pose[k]; pose[k] = s.final_pose[k]; }
        return s.final_status;
    });
}

int32_t lvi_scan_replay_enqueue(lvi_lidar* h, const void* d_pts, int32_t n_raw, const float pose_init[6], void* d_record, int32_t rebuild_map)
{
    if (!h || !pose_init || (n_raw > 0 && !d_pts)) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (n_raw > h->cur().raw_cap) return fail(LVI_ERR_CAPACITY, "n_raw exceeds max_raw_points");
    if (rebuild_map && !h->d.have_map_raw) return fail(LVI_ERR_STATE, "no map uploaded");
    if (h->sel != 0) return fail(LVI_ERR_STATE, "lvi_scan_replay_enqueue runs on slot 0; batches go through lvi_scan_batch_run");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        join_map(d);
        // per-call inputs go in eagerly; the captured graph only reads fixed buffers of the handle
        d.raw_bound = nullptr;
        d.n_raw = n_raw > 0 ? n_raw - 1 : 0;                     // moveFromCustomMsg drops the final point
        if (d.n_raw) LVI_HIP(hipMemcpyAsync(d.raw, d_pts, sizeof(lvi_livox_pt) * (size_t)d.n_raw, hipMemcpyDeviceToDevice, d.ctx.stream));
        LVI_HIP(hipMemsetAsync(d.d_status, 0, sizeof(int), d.ctx.stream));
        set_pose_init(d, pose_init);
        d.have_raw = true;
        auto run_stages = [&]() {
            if (rebuild_map) { stage_map_build(d); d.have_map = true; }
            stage_organize(d); d.have_org = true;
            stage_extract(d); d.have_feat = true;
            stage_downsample(d); d.have_ds = true;
            stage_scan_match_enqueue(d, nullptr, nullptr);
        };
        if (d.prof.on || d.dk_on) {
            run_stages();                                         // per-kernel HIP events need eager launches; the deskew table's time origin is a kernel argument
        } else {
            // everything the captured launch sequence froze: sizes, whether a map index exists (icp_init's argument) and the
            // realisation AUTO picked for each voxel plan (unknown = sorted on a plan's first batch, binned afterwards)
            const std::array<int, 8> key = {d.n_raw, d.n_map_corner, d.n_map_surf, (rebuild_map ? 1 : 0) | (d.voxMap.bbox_cached ? 2 : 0) | (d.voxMap.hist_cached ? 4 : 0), (d.have_map || rebuild_map) ? 1 : 0,
                                            voxel_resolve_mode(d.voxRing), voxel_resolve_mode(d.voxScan), voxel_resolve_mode(d.voxMap)};
            const bool stale = !d.graphExec || key != d.graph_key;
            if (stale) {
                if (d.graphExec) { (void)hipGraphExecDestroy(d.graphExec); d.graphExec = nullptr; }
                hipGraph_t graph = nullptr;
                LVI_HIP(hipStreamBeginCapture(d.ctx.stream, hipStreamCaptureModeThreadLocal));
                try { run_stages(); } catch (...) { (void)hipStreamEndCapture(d.ctx.stream, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
                LVI_HIP(hipStreamEndCapture(d.ctx.stream, &graph));
                LVI_HIP(hipGraphInstantiate(&d.graphExec, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                d.graph_key = key;
            }
            d.have_map = d.have_map || rebuild_map; d.have_org = d.have_feat = d.have_ds = true;
            LVI_HIP(hipGraphLaunch(d.graphExec, d.ctx.stream));
        }
        if (d_record) LVI_HIP(hipMemcpyAsync(d_record, &d.icp->record, sizeof(lvi_pose_record), hipMemcpyDeviceToDevice, d.ctx.stream));
        h->have_icp_host = false;
        return LVI_OK;
    });
}

// ---- batched form: S scans side by side in one launch sequence ------------------------------------
static int32_t batch_check(lvi_lidar* h, int32_t n)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    if (n < 1 || n > (int32_t)h->slots.size()) return fail(LVI_ERR_CAPACITY, "n_scans exceeds lvi_lidar_params.batch_scans");
    return LVI_OK;
}
int32_t lvi_batch_select(lvi_lidar* h, int32_t slot)
{
    if (!h || slot < 0 || slot >= (int32_t)h->slots.size()) return fail(LVI_ERR_INVALID_ARG, "bad slot");
    h->sel = slot;
    return LVI_OK;
}
int32_t lvi_scan_batch_bind_device(lvi_lidar* h, int32_t n_scans, const void* const* d_pts, const int32_t* n_raw)
{
    int32_t st = batch_check(h, n_scans); if (st) return st;
    if (!d_pts || !n_raw) return fail(LVI_ERR_INVALID_ARG, "null argument");
    for (int z = 0; z < n_scans; z++) {
        if (n_raw[z] < 0 || (n_raw[z] > 0 && !d_pts[z])) return fail(LVI_ERR_INVALID_ARG, "bad scan");
        if (n_raw[z] > h->d.raw_cap) return fail(LVI_ERR_CAPACITY, "n_raw exceeds max_raw_points");
    }
    for (int z = 0; z < n_scans; z++) {
        LidarDev& q = *h->slots[z];
        q.raw_bound = (const lvi_livox_pt*)d_pts[z];
        q.n_raw = n_raw[z] > 0 ? n_raw[z] - 1 : 0;                  // moveFromCustomMsg drops the final point (imageProjection.cpp:249)
        q.have_raw = true; q.have_org = q.have_feat = q.have_ds = false;
    }
    return LVI_OK;
}
int32_t lvi_scan_batch_upload(lvi_lidar* h, int32_t n_scans, const lvi_livox_pt* const* pts, const int32_t* n_raw)
{
    int32_t st = batch_check(h, n_scans); if (st) return st;
    if (!pts || !n_raw) return fail(LVI_ERR_INVALID_ARG, "null argument");
    for (int z = 0; z < n_scans; z++) {
        if (n_raw[z] < 0 || (n_raw[z] > 0 && !pts[z])) return fail(LVI_ERR_INVALID_ARG, "bad scan");
        if (n_raw[z] > h->d.raw_cap) return fail(LVI_ERR_CAPACITY, "n_raw exceeds max_raw_points");
    }
    return guarded(h, [&]() -> int32_t {
        for (int z = 0; z < n_scans; z++) {
            LidarDev& q = *h->slots[z];
            q.raw_bound = nullptr;
            q.n_raw = n_raw[z] > 0 ? n_raw[z] - 1 : 0;
            const int slot = (q.raw_slot ^= 1);                      // pinned staging ring, as lvi_scan_upload
            LVI_HIP(hipEventSynchronize(q.ev_raw[slot]));
            std::memcpy(q.h_raw[slot], pts[z], sizeof(lvi_livox_pt) * (size_t)q.n_raw);
            if (q.n_raw) LVI_HIP(hipMemcpyAsync(q.raw, q.h_raw[slot], sizeof(lvi_livox_pt) * (size_t)q.n_raw, hipMemcpyHostToDevice, q.ctx.stream));
            LVI_HIP(hipEventRecord(q.ev_raw[slot], q.ctx.stream));
            q.have_raw = true; q.have_org = q.have_feat = q.have_ds = false;
        }
        return LVI_OK;
    });
}
int32_t lvi_scan_batch_run(lvi_lidar* h, int32_t n_scans, const float* pose_init, void* d_records, int32_t rebuild_map)
{
    int32_t st = batch_check(h, n_scans); if (st) return st;
    if (!pose_init) return fail(LVI_ERR_INVALID_ARG, "null argument");
    for (int z = 0; z < n_scans; z++) if (!h->slots[z]->have_raw) return fail(LVI_ERR_STATE, "no scan bound / uploaded for a slot");
    if (rebuild_map && !h->d.have_map_raw) return fail(LVI_ERR_STATE, "no map uploaded");
    return guarded(h, [&]() -> int32_t {
        const Slots sl{h->slots.data(), n_scans};
        set_pose_init(sl, pose_init, true);
        if (rebuild_map) stage_map_build(sl);
        stage_organize(sl);
        stage_extract(sl);
        stage_downsample(sl);
        for (int z = 0; z < n_scans; z++) { LidarDev& q = sl[z]; q.have_org = q.have_feat = q.have_ds = true; }
        stage_scan_match_enqueue(sl, nullptr, d_records);
        h->have_icp_host = false;
        return LVI_OK;
    });
}
int32_t lvi_scan_batch_get_records(lvi_lidar* h, int32_t n_scans, lvi_pose_record* out)
{
    int32_t st = batch_check(h, n_scans); if (st) return st;
    if (!out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    return guarded(h, [&]() -> int32_t {
        for (int z = 0; z < n_scans; z++) d2h(h->d, out + z, &h->slots[z]->icp->record, 1);
        sync(h->d);
        return LVI_OK;                                               // device errors travel in each record's status
    });
}

// ---- fetch ---------------------------------------------------------------------------------------
int32_t lvi_get_scan_info(lvi_lidar* h, lvi_scan_info* out)
{
    if (!h || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->cur().have_org) return fail(LVI_ERR_STATE, "scan not organised");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        const int n = read_int(d, d.d_n);
        out->n = n;
        if (out->capacity < n) return fail(LVI_ERR_CAPACITY, "scan_info capacity too small");
        d2h(d, out->start_ring_index, d.startR, (size_t)d.P.N_SCAN); d2h(d, out->end_ring_index, d.endR, (size_t)d.P.N_SCAN);
        d2h(d, out->point_col_ind, d.col, (size_t)n); d2h(d, out->point_range, d.range, (size_t)n); d2h(d, out->cloud_deskewed, d.pts, (size_t)n);
        sync(d);
        return LVI_OK;
    });
}
int32_t lvi_get_features(lvi_lidar* h, lvi_cloud* corner, lvi_cloud* surf)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->cur().have_feat) return fail(LVI_ERR_STATE, "features not extracted");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        int32_t st = check_dev_status(d); if (st) return st;
        const Counts c = read_counts(d);
        st = fetch_cloud(d, d.corner, c.ncorner, corner); if (st) return st;
        return fetch_cloud(d, d.surf, c.nsurf, surf);
    });
}
int32_t lvi_get_scan_ds(lvi_lidar* h, lvi_cloud* c0, lvi_cloud* c1)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->cur().have_ds) return fail(LVI_ERR_STATE, "scan not downsampled");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        const Counts c = read_counts(d);
        int32_t st = fetch_cloud(d, d.cornerDS, c.ncds, c0); if (st) return st;
        return fetch_cloud(d, d.surfDS, c.nsds, c1);
    });
}
int32_t lvi_get_map_ds(lvi_lidar* h, lvi_cloud* c0, lvi_cloud* c1)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->cur().have_map) return fail(LVI_ERR_STATE, "map not built");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        int32_t st = check_dev_status(d); if (st) return st;
        const Counts c = read_counts(d);
        st = fetch_cloud(d, d.mapCornerDS, c.mcds, c0); if (st) return st;
        return fetch_cloud(d, d.mapSurfDS, c.msds, c1);
    });
}
int32_t lvi_get_counts(lvi_lidar* h, int32_t counts[8])
{
    if (!h || !counts) return fail(LVI_ERR_INVALID_ARG, "null argument");
    return guarded(h, [&]() -> int32_t {
        const Counts c = read_counts(h->cur());
        counts[0] = c.n; counts[1] = c.ncorner; counts[2] = c.nsurf; counts[3] = c.ncds; counts[4] = c.nsds; counts[5] = c.mcds; counts[6] = c.msds; counts[7] = 0;
        return LVI_OK;
    });
}

int32_t lvi_get_pose_record(lvi_lidar* h, lvi_pose_record* out)
{
    if (!h || !out) return fail(LVI_ERR_INVALID_ARG, "null argument");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        d2h(d, out, &d.icp->record, 1);
        sync(d);
        return check_dev_status(d);
    });
}

// ---- one-call forms ------------------------------------------------------------------------------
// ---- f-4 -------------------------------------------------------------------------------------------------
static int32_t kf_reserve(lvi_lidar* h, int nc, int ns)
{
    LidarDev& d = h->d;
    if ((int)d.kf_pose.size() >= d.P.max_keyframes || (long long)d.kf_pool_used + nc + ns > (long long)d.kf_pool_cap)
        return fail(LVI_ERR_CAPACITY, "keyframe store full");
    return LVI_OK;
}
static int32_t kf_commit(lvi_lidar* h, int nc, int ns, const float pose[6], int32_t* index_out)
{
    LidarDev& d = h->d;
    d.kf_off_c.push_back(d.kf_pool_used); d.kf_n_c.push_back(nc);
    d.kf_off_s.push_back(d.kf_pool_used + nc); d.kf_n_s.push_back(ns);
    d.kf_pool_used += nc + ns;
    d.kf_pose.push_back({pose[0], pose[1], pose[2], pose[3], pose[4], pose[5]});
    if (index_out) *index_out = (int32_t)d.kf_pose.size() - 1;
    return LVI_OK;
}
int32_t lvi_keyframe_add(lvi_lidar* h, const lvi_pt* corner, int32_t nc, const lvi_pt* surf, int32_t ns, const float pose[6], int32_t* index_out)
{
    if (!h || nc < 0 || ns < 0 || (nc > 0 && !corner) || (ns > 0 && !surf) || !pose) return fail(LVI_ERR_INVALID_ARG, "bad keyframe arguments");
    int32_t st = kf_reserve(h, nc, ns); if (st) return st;
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        h2d(d, d.kfPool + d.kf_pool_used, corner, (size_t)nc);
        h2d(d, d.kfPool + d.kf_pool_used + nc, surf, (size_t)ns);
        sync(d);
        return kf_commit(h, nc, ns, pose, index_out);
    });
}
int32_t lvi_keyframe_add_current(lvi_lidar* h, const float pose[6], int32_t* index_out)
{
    if (!h || !pose) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (!h->cur().have_ds) return fail(LVI_ERR_STATE, "scan not downsampled");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;                                          // the store
        LidarDev& q = h->cur();                                      // the scan (selected batch slot)
        int nq[3] = {0, 0, 0};
        d2h(q, nq, q.voxScan.d_nout, 3); sync(q);                    // a keyframe is saved every ~1 m of motion: one 12-byte read
        int32_t st = kf_reserve(h, nq[0], nq[1]); if (st) return st;
        if (nq[0]) LVI_HIP(hipMemcpyAsync(d.kfPool + d.kf_pool_used, q.cornerDS, sizeof(lvi_pt) * (size_t)nq[0], hipMemcpyDeviceToDevice, d.ctx.stream));
        if (nq[1]) LVI_HIP(hipMemcpyAsync(d.kfPool + d.kf_pool_used + nq[0], q.surfDS, sizeof(lvi_pt) * (size_t)nq[1], hipMemcpyDeviceToDevice, d.ctx.stream));
        mark_map_deps(d);                                             // a later map update reads these clouds on the second stream
        return kf_commit(h, nq[0], nq[1], pose, index_out);
    });
}
int32_t lvi_keyframe_set_pose(lvi_lidar* h, int32_t index, const float pose[6])
{
    if (!h || !pose || index < 0 || index >= (int32_t)h->d.kf_pose.size()) return fail(LVI_ERR_INVALID_ARG, "bad keyframe index");
    for (int k = 0; k < 6; k++) h->d.kf_pose[index][k] = pose[k];
    return LVI_OK;
}
int32_t lvi_keyframe_count(lvi_lidar* h, int32_t* n_keyframes, int32_t* n_points)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    if (n_keyframes) *n_keyframes = (int32_t)h->d.kf_pose.size();
    if (n_points) *n_points = h->d.kf_pool_used;
    return LVI_OK;
}
int32_t lvi_keyframes_clear(lvi_lidar* h)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        join_map(d); sync(d);                                        // an assembly in flight still reads the pool
        d.kf_off_c.clear(); d.kf_n_c.clear(); d.kf_off_s.clear(); d.kf_n_s.clear(); d.kf_pose.clear(); d.kf_pool_used = 0;
        d.inc_ready = false; d.inc_mult.clear(); d.inc_pose.clear();
        return LVI_OK;
    });
}
int32_t lvi_map_assemble(lvi_lidar* h, const int32_t* key_indices, int32_t n_keys)
{
    if (!h || n_keys < 0 || (n_keys > 0 && !key_indices)) return fail(LVI_ERR_INVALID_ARG, "bad key list");
    if (2 * n_keys > h->d.kf_seg_cap) return fail(LVI_ERR_CAPACITY, "key list longer than the assembly table");
    long long tc = 0, ts = 0;
    for (int i = 0; i < n_keys; i++) {
        const int k = key_indices[i];
        if (k < 0 || k >= (int)h->d.kf_pose.size()) return fail(LVI_ERR_INVALID_ARG, "key index out of range");
        tc += h->d.kf_n_c[k]; ts += h->d.kf_n_s[k];
    }
    if (tc > h->d.map_cap || ts > h->d.map_cap) return fail(LVI_ERR_CAPACITY, "map exceeds max_map_points");
    if (h->shared_by > 0) return fail(LVI_ERR_STATE, "other handles read this handle's map (lvi_map_share)");
    return guarded(h, [&]() -> int32_t {
        unshare_map(h);
        drop_graph(h);
        stage_map_assemble(h->d, key_indices, n_keys);                                      // extractCloud's fuse loop into the raw map (slot 0)
        stage_map_build(Slots{h->slots.data(), (int)h->slots.size()});                      // + the two VoxelGrids and the index, per slot
        return LVI_OK;
    });
}

int32_t lvi_map_update(lvi_lidar* h, const int32_t* key_indices, int32_t n_keys)
{
    if (!h || n_keys < 0 || (n_keys > 0 && !key_indices)) return fail(LVI_ERR_INVALID_ARG, "bad key list");
    for (int i = 0; i < n_keys; i++)
        if (key_indices[i] < 0 || key_indices[i] >= (int)h->d.kf_pose.size()) return fail(LVI_ERR_INVALID_ARG, "key index out of range");
    if (h->shared_by > 0) return fail(LVI_ERR_STATE, "other handles read this handle's map (lvi_map_share)");
    bool done = false;
    int32_t st = guarded(h, [&]() -> int32_t {
        drop_graph(h);
        done = h->slots.size() == 1 && stage_map_update(h->d, key_indices, n_keys);
        return LVI_OK;
    });
    if (st) return st;
    if (done) return LVI_OK;
    return lvi_map_assemble(h, key_indices, n_keys);         // batch handle, range / overflow rule / table full: the full path
}

int32_t lvi_scan_set_deskew(lvi_lidar* h, const lvi_deskew_info* info)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    h->cur().dk_on = false;
    if (!info || !info->imu_available) return LVI_OK;
    if (info->imu_pointer_cur < 1 || info->imu_pointer_cur >= LVI_DESKEW_MAX_IMU || !info->imu_time || !info->imu_rot_x || !info->imu_rot_y || !info->imu_rot_z)
        return fail(LVI_ERR_INVALID_ARG, "bad deskew table");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        const size_t m = (size_t)info->imu_pointer_cur + 1;
        const double* src[4] = {info->imu_time, info->imu_rot_x, info->imu_rot_y, info->imu_rot_z};
        for (int k = 0; k < 4; k++) h2d(d, d.d_dk + (size_t)k * LVI_DESKEW_MAX_IMU, src[k], m);
        sync(d);                                            // the caller's arrays may go away after the call
        d.dk_cur = info->imu_pointer_cur; d.dk_t0 = info->time_scan_cur; d.dk_on = true;
        return LVI_OK;
    });
}

int32_t lvi_organize_scan_deskew(lvi_lidar* h, const lvi_livox_pt* pts, int32_t n_raw, const lvi_deskew_info* info, lvi_scan_info* out)
{
    int32_t st = lvi_scan_set_deskew(h, info); if (st) return st;
    st = lvi_scan_upload(h, pts, n_raw); if (st) return st;
    st = lvi_scan_organize(h); if (st) return st;
    return lvi_get_scan_info(h, out);
}

int32_t lvi_organize_scan(lvi_lidar* h, const lvi_livox_pt* pts, int32_t n_raw, lvi_scan_info* out)
{
    int32_t st = lvi_scan_set_deskew(h, nullptr); if (st) return st;
    st = lvi_scan_upload(h, pts, n_raw); if (st) return st;
    st = lvi_scan_organize(h); if (st) return st;
    return lvi_get_scan_info(h, out);
}

int32_t lvi_extract_features(lvi_lidar* h, const lvi_scan_info* in, lvi_cloud* corner, lvi_cloud* surf)
{
    if (!h || !in) return fail(LVI_ERR_INVALID_ARG, "null argument");
    if (in->n < 0 || in->n > h->cur().ext_cap) return fail(LVI_ERR_CAPACITY, "scan_info.n exceeds capacity");
    int32_t st = guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        const int n = in->n, NS = d.P.N_SCAN;
        // ring bases: start_ring_index[r] = count_r + 4 (imageProjection.cpp:630)
        std::vector<int> base(NS + 1);
        for (int r = 0; r < NS; r++) base[r] = in->start_ring_index[r] - 4;
        base[NS] = n;
        for (int r = 0; r < NS; r++) if (base[r] < 0 || base[r] > base[r + 1]) return fail(LVI_ERR_INVALID_ARG, "inconsistent ring indices");
        h2d(d, d.pts, in->cloud_deskewed, (size_t)n); h2d(d, d.range, in->point_range, (size_t)n); h2d(d, d.col, in->point_col_ind, (size_t)n);
        h2d(d, d.startR, in->start_ring_index, (size_t)NS); h2d(d, d.endR, in->end_ring_index, (size_t)NS);
        d.raw_bound = nullptr;
        h2d(d, d.ringBase, base.data(), (size_t)NS + 1); h2d(d, d.d_n, &n, 1);
        LVI_HIP(hipMemsetAsync(d.d_status, 0, sizeof(int), d.ctx.stream));
        sync(d);
        d.n_raw = n; d.have_org = true;
        return LVI_OK;
    });
    if (st) return st;
    st = lvi_scan_extract(h); if (st) return st;
    return lvi_get_features(h, corner, surf);
}

int32_t lvi_voxel_downsample(lvi_lidar* h, const lvi_pt* in, int32_t n, float leaf, lvi_pt* out, int32_t out_capacity, int32_t* n_out)
{
    if (!h || n < 0 || (n > 0 && !in) || !(leaf > 0.f) || !n_out) return fail(LVI_ERR_INVALID_ARG, "bad voxel arguments");
    if (n > h->d.voxGen.seg_cap) return fail(LVI_ERR_CAPACITY, "n exceeds capacity");
    const bool no_tiny = getenv("LVI_VOX_NO_TINY") != nullptr;                // tests: the general path for every size
    if (n > 0 && n <= VOX_TINY && !no_tiny) return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        if (!h->vt_in) {
            LVI_HIP(hipHostMalloc((void**)&h->vt_in, sizeof(lvi_pt) * VOX_TINY, hipHostMallocDefault));
            LVI_HIP(hipHostMalloc((void**)&h->vt_out, sizeof(lvi_pt) * VOX_TINY, hipHostMallocDefault));
            LVI_HIP(hipHostMalloc((void**)&h->vt_hdr, 64, hipHostMallocDefault));
            LVI_HIP(hipHostMalloc((void**)&h->vt_cells, sizeof(int) * VOX_TINY, hipHostMallocDefault));
            LVI_HIP(hipHostMalloc((void**)&h->vt_counts, sizeof(int) * VOX_TINY, hipHostMallocDefault));
            LVI_HIP(hipHostMalloc((void**)&h->vt_keys, sizeof(int) * VOX_TINY, hipHostMallocDefault));
        }
        // On the handle's SECOND stream: the call returns its result, so it orders with nothing else — and the node calls it (key-pose grid,
        // extractNearby) right after it enqueued the scan's voxel grids on the main stream: there the wait below would be a wait for all of
        // them, and the local-map update that follows would start when the scan-side stages are over instead of beside them.
        // (The staging block is single; the previous call waited for its kernel.)
        std::memcpy(h->vt_in, in, sizeof(lvi_pt) * (size_t)n);
        voxel_tiny(d.ctx2, h->vt_in, n, leaf, d.voxGen.seg_cap, d.voxGen.bin_pts, d.voxGen.bin_max, h->vt_out, h->vt_hdr, h->vt_cells, h->vt_counts, h->vt_keys);
        LVI_HIP(hipStreamSynchronize(d.ctx2.stream));
        const int m = h->vt_hdr[0];
        *n_out = m;
        const int fetch = std::min(m, out_capacity);
        if (fetch > 0) std::memcpy(out, h->vt_out, sizeof(lvi_pt) * (size_t)fetch);
        h->vdbg_pending = false;
        if (h->vt_hdr[1]) { h->vkeys.clear(); h->vcells.clear(); h->vcounts.clear(); }      // overflow rule: no voxels to show (as the general path's views)
        else { h->vkeys.assign(h->vt_keys, h->vt_keys + n); h->vcells.assign(h->vt_cells, h->vt_cells + m); h->vcounts.assign(h->vt_counts, h->vt_counts + m); }
        if (m > out_capacity) return fail(LVI_ERR_CAPACITY, "voxel output capacity too small");
        return LVI_OK;
    });
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        h2d(d, d.genIn, in, (size_t)n);
        if (!d.gen_static_set || d.gen_leaf != leaf) {               // (the segment table is uploaded once per leaf size: the node calls this with one leaf for every scan)
            VoxSegStatic st{d.genIn, nullptr, d.genOut, leaf};
            d.voxGen.set_static(d.ctx, &st);
            d.gen_static_set = true; d.gen_leaf = leaf;
        }
        d.voxGen.n_host[0] = n; d.voxGen.use_n_host = true;          // the length travels as a kernel argument
        voxel_downsample_batch(d.ctx, d.voxGen, "gen", n);
        // ONE wait: the count and (at most the caller's capacity of) the output travel together
        const int fetch = std::min(n, out_capacity);
        int m = 0;
        d2h(d, &m, d.voxGen.d_nout, 1);
        if (fetch > 0) d2h(d, out, d.genOut, (size_t)fetch);
        sync(d);
        *n_out = m;
        h->vdbg_pending = true; h->vdbg_n = n;                       // LVI_DBG_VOXEL_*: fetched when asked for (until the next call that uses the generic buffers)
        if (m > out_capacity) return fail(LVI_ERR_CAPACITY, "voxel output capacity too small");
        return LVI_OK;
    });
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
    if (nc > h->cur().ext_cap || ns > h->cur().ext_cap || (long long)nc + ns > h->cur().ext_cap) return fail(LVI_ERR_CAPACITY, "feature clouds exceed capacity (corner + surf <= N_SCAN * Horizon_SCAN)");
    int32_t st = guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        h2d(d, d.corner, corner, (size_t)nc); h2d(d, d.surf, surf, (size_t)ns);
        h2d(d, d.d_ncorner, &nc, 1); h2d(d, d.voxRing.d_nout + d.P.N_SCAN, &ns, 1);
        sync(d);
        d.n_raw = nc + ns; d.have_feat = true;
        return LVI_OK;
    });
    if (st) return st;
    st = lvi_scan_downsample(h); if (st) return st;
    return lvi_scan_match(h, imu, pose, out);
}

int32_t lvi_transform_cloud(lvi_lidar* h, const lvi_pt* in, int32_t n, const float pose6[6], lvi_pt* out)
{
    if (!h || n < 0 || (n > 0 && (!in || !out)) || !pose6) return fail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (n > h->d.voxGen.seg_cap) return fail(LVI_ERR_CAPACITY, "n exceeds capacity");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->d;
        h->vdbg_pending = false;
        h2d(d, d.genIn, in, (size_t)n);
        transform_cloud(d, d.genIn, n, pose6, d.genOut);
        d2h(d, out, d.genOut, (size_t)n);
        sync(d);
        return LVI_OK;
    });
}

// ---- inspection ------------------------------------------------------------------------------------
int32_t lvi_debug_get(lvi_lidar* h, int32_t what, void* dst, int64_t cap, int64_t* n_bytes)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        auto need_feat = [&]() { return d.have_feat && d.have_org; };
        switch (what) {
            case LVI_DBG_CURVATURE: {
                if (!need_feat()) return fail(LVI_ERR_STATE, "stage not run");
                const int n = read_int(d, d.d_n);
                std::vector<float> v(n); d2h(d, v.data(), d.curv, (size_t)n); sync(d);
                return dbg_out(v, 0, dst, cap, n_bytes);
            }
            case LVI_DBG_PICKED_OCCL: case LVI_DBG_PICKED_FINAL: case LVI_DBG_LABEL: {
                if (!need_feat()) return fail(LVI_ERR_STATE, "stage not run");
                const int n = read_int(d, d.d_n);
                std::vector<int8_t> b(n);
                const void* src = what == LVI_DBG_PICKED_OCCL ? (const void*)d.picked_occl : what == LVI_DBG_PICKED_FINAL ? (const void*)d.picked : (const void*)d.label;
                if (n) LVI_HIP(hipMemcpyAsync(b.data(), src, (size_t)n, hipMemcpyDeviceToHost, d.ctx.stream));
                sync(d);
                std::vector<int32_t> v(n);
                for (int i = 0; i < n; i++) v[i] = b[i];
                return dbg_out(v, 0, dst, cap, n_bytes);
            }
            case LVI_DBG_CORNER_INDEX: {
                if (!need_feat()) return fail(LVI_ERR_STATE, "stage not run");
                const int n = read_int(d, d.d_ncorner);
                std::vector<int32_t> v(n); d2h(d, v.data(), d.corner_idx, (size_t)n); sync(d);
                return dbg_out(v, 0, dst, cap, n_bytes);
            }
            case LVI_DBG_MAP_CORNER_RAW: case LVI_DBG_MAP_SURF_RAW: {
                if (!h->d.have_map_raw) return fail(LVI_ERR_STATE, "no map");
                const bool sf = what == LVI_DBG_MAP_SURF_RAW;
                const int n = sf ? h->d.n_map_surf : h->d.n_map_corner;
                std::vector<lvi_pt> v((size_t)n); d2h(d, v.data(), sf ? d.mapSurfRaw : d.mapCornerRaw, (size_t)n); sync(d);
                return dbg_out(v, 0, dst, cap, n_bytes);
            }
            case LVI_DBG_ICP_CYCLES: {
                std::vector<long long> v(16); d2h(d, v.data(), d.d_icp_cycles, 16); sync(d);
                return dbg_out(v, 0, dst, cap, n_bytes);
            }
            case LVI_DBG_FEAT_CYCLES: {
                if (!need_feat()) return fail(LVI_ERR_STATE, "stage not run");
                std::vector<long long> v(8); d2h(d, v.data(), d.d_feat_cycles, 8); sync(d);
                return dbg_out(v, 0, dst, cap, n_bytes);
            }
            case LVI_DBG_VOXEL_KEYS: case LVI_DBG_VOXEL_CELLS: case LVI_DBG_VOXEL_COUNTS:
                if (h->vdbg_pending) { voxel_debug_fetch(h->d.ctx, h->d.voxGen, h->vdbg_n, h->vkeys, h->vcells, h->vcounts); h->vdbg_pending = false; }
                return dbg_out(what == LVI_DBG_VOXEL_KEYS ? h->vkeys : (what == LVI_DBG_VOXEL_CELLS ? h->vcells : h->vcounts), 0, dst, cap, n_bytes);
            case LVI_DBG_ICP_JTJ: {
                if (!h->have_icp_host) return fail(LVI_ERR_STATE, "scan_match not run");
                if (!h->icp_host_full) { d2h(d, d.h_icp, d.icp, 1); sync(d); h->icp_host_full = true; }
                std::vector<float> v(d.h_icp->jtj, d.h_icp->jtj + 27 * d.h_icp->iters);
                return dbg_out(v, 0, dst, cap, n_bytes);
            }
            case LVI_DBG_ICP_POSE_TRACE: {
                if (!h->have_icp_host) return fail(LVI_ERR_STATE, "scan_match not run");
                if (!h->icp_host_full) { d2h(d, d.h_icp, d.icp, 1); sync(d); h->icp_host_full = true; }
                const int rows = d.h_icp->status == LVI_OK || d.h_icp->status == LVI_TOO_FEW_CORRESPONDENCES ? d.h_icp->iters + 1 : 0;
                std::vector<float> v(d.h_icp->pose_trace, d.h_icp->pose_trace + 6 * rows);
                return dbg_out(v, 0, dst, cap, n_bytes);
            }
            default: return fail(LVI_ERR_INVALID_ARG, "unknown debug item");
        }
    });
}

int32_t lvi_debug_knn(lvi_lidar* h, int32_t which, const lvi_pt* queries, int32_t nq, int32_t* idx, float* sqd)
{
    if (!h || !queries || !idx || !sqd || nq < 0 || which < 0 || which > 1) return fail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (!h->cur().have_map) return fail(LVI_ERR_STATE, "map not built");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        struct Scratch {                                   // freed on every way out, a throwing LVI_HIP included
            void* p[3] = {nullptr, nullptr, nullptr};
            ~Scratch() { for (void* q : p) if (q) (void)hipFree(q); }
        } sc;
        const size_t m = (size_t)std::max(nq, 1);
        LVI_HIP(hipMalloc(&sc.p[0], sizeof(lvi_pt) * m));
        LVI_HIP(hipMalloc(&sc.p[1], sizeof(int) * 5 * m));
        LVI_HIP(hipMalloc(&sc.p[2], sizeof(float) * 5 * m));
        lvi_pt* dq = (lvi_pt*)sc.p[0]; int* di = (int*)sc.p[1]; float* dd = (float*)sc.p[2];
        h2d(d, dq, queries, (size_t)nq);
        debug_knn(d, which, dq, nq, di, dd);
        d2h(d, idx, di, (size_t)nq * 5); d2h(d, sqd, dd, (size_t)nq * 5);
        sync(d);
        return check_dev_status(d);
    });
}

int32_t lvi_debug_residuals(lvi_lidar* h, int32_t which, const float pose[6], lvi_pt* coeff, uint8_t* flag, int32_t capacity, int32_t* n)
{
    if (!h || !pose || !coeff || !flag || !n || which < 0 || which > 1) return fail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (!h->cur().have_map || !h->cur().have_ds) return fail(LVI_ERR_STATE, "map or scan DS missing");
    return guarded(h, [&]() -> int32_t {
        LidarDev& d = h->cur();
        int nq[3];
        d2h(d, nq, d.voxScan.d_nout, 3); sync(d);
        *n = nq[which];
        if (capacity < *n) return fail(LVI_ERR_CAPACITY, "capacity too small");
        debug_residuals(d, which, pose);
        d2h(d, coeff, d.coeff, (size_t)*n);
        if (*n) LVI_HIP(hipMemcpyAsync(flag, d.flag, (size_t)*n, hipMemcpyDeviceToHost, d.ctx.stream));
        sync(d);
        return LVI_OK;
    });
}

// ---- kernel timing -----------------------------------------------------------------------------------
int32_t lvi_prof_enable(lvi_lidar* h, int32_t on)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    return guarded(h, [&]() -> int32_t { sync(h->d); h->d.prof.collect(); h->d.prof.on = on != 0; return LVI_OK; });
}
int32_t lvi_prof_reset(lvi_lidar* h)
{
    if (!h) return fail(LVI_ERR_INVALID_ARG, "null handle");
    return guarded(h, [&]() -> int32_t { sync(h->d); h->d.prof.reset(); return LVI_OK; });
}
int32_t lvi_prof_read(lvi_lidar* h, lvi_kernel_stat* stats, int32_t capacity, int32_t* n)
{
    if (!h || !n) return fail(LVI_ERR_INVALID_ARG, "null argument");
    return guarded(h, [&]() -> int32_t {
        Profiler& p = h->d.prof;
        sync(h->d); p.collect();
        int k = 0;
        for (size_t i = 0; i < p.names.size(); i++) {
            if (!p.launches[i]) continue;
            if (k < capacity && stats) {
                memset(&stats[k], 0, sizeof(stats[k]));
                strncpy(stats[k].name, p.names[i].c_str(), sizeof(stats[k].name) - 1);
                stats[k].launches = p.launches[i]; stats[k].total_ms = p.total_ms[i]; stats[k].bytes_alg = p.bytes[i];
            }
            k++;
        }
        *n = std::min(k, capacity);
        return LVI_OK;
    });
}

}  // extern "C"
