// Scan-side kernels of the lidar path for gfx950 (SURVEY §8 a-0 … a-4):
//   organise   ImageProjection::projectPointCloud + cloudExtraction  (imageProjection.cpp:570-647)
//   smooth     FeatureExtraction::calculateSmoothness + markOccludedPoints (featureExtraction.cpp:87-148)
//   sectors    FeatureExtraction::extractFeatures greedy selection   (featureExtraction.cpp:158-237)
//   per-ring / per-scan VoxelGrid                                     (:239-243, mapOptimization.cpp:987-999)
// All counts stay in device memory; the host only enqueues.
#include "lvi_lidar.hpp"

namespace lvi {

namespace {

// ---------------------------------------------------------------------------------------------
// a-0.  The reference walks the raw points serially: range gate, ring gate, Livox column =
// running per-ring counter, scatter to fullCloud[col + ring*H], then a row-major compaction.
// Because the column is a dense per-ring counter, the result is exactly a STABLE partition of the
// surviving points by ring, truncated to Horizon_SCAN per ring — done here as one counting-sort
// pass: classify+count per tile, scan, ranked scatter (wave64 ballot match on the ring id).
// ---------------------------------------------------------------------------------------------
struct OrgArgs {
    const lvi_livox_pt* raw; int n_raw;
    int* blockCnt; int nblk;
    int *ringBase, *startR, *endR, *d_n;
    lvi_pt* pts; float* range; int* col;
    int N_SCAN, H, downsampleRate; float minRange, maxRange;
    int* d_status;
};

__device__ __forceinline__ int org_classify(const OrgArgs& a, const lvi_livox_pt& p, float* range_out)
{
    const float r = sqrt_rn(add_rn(add_rn(mul_rn(p.x, p.x), mul_rn(p.y, p.y)), mul_rn(p.z, p.z)));   // utility.h:403-406
    *range_out = r;
    if (r < a.minRange || r > a.maxRange) return -1;            // imageProjection.cpp:583
    const int row = (int)p.line;                                 // :586 (ring = line, :257)
    if (row < 0 || row >= a.N_SCAN) return -1;
    if (row % a.downsampleRate != 0) return -1;                  // :590
    return row;
}

__global__ __launch_bounds__(256) void org_count_kernel(OrgArgs a)
{
    __shared__ int cnt[MAX_N_SCAN];
    if (threadIdx.x < MAX_N_SCAN) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * ORG_TILE;
#pragma unroll
    for (int j = 0; j < ORG_TILE / 256; j++) {
        const int i = base + j * 256 + threadIdx.x;
        if (i < a.n_raw) {
            float r;
            const int ring = org_classify(a, a.raw[i], &r);
            if (ring >= 0) atomicAdd(&cnt[ring], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < a.N_SCAN) a.blockCnt[threadIdx.x * a.nblk + blockIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(256) void org_scan_kernel(OrgArgs a)
{
    __shared__ int ws[8];
    __shared__ int total[MAX_N_SCAN];
    const int nt = (a.n_raw + ORG_TILE - 1) / ORG_TILE;
    for (int r = 0; r < a.N_SCAN; r++) {
        int* row = a.blockCnt + r * a.nblk;
        int carry = 0;
        for (int c = 0; c < nt; c += 256) {
            const int i = c + threadIdx.x;
            const int v = (i < nt) ? row[i] : 0;
            int tot;
            const int ex = block_excl_scan<256>(v, ws, &tot);
            if (i < nt) row[i] = carry + ex;
            carry += tot;
        }
        if (threadIdx.x == 0) total[r] = carry;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int count = 0;
        for (int r = 0; r < a.N_SCAN; r++) {                    // cloudExtraction :628-646
            a.ringBase[r] = count;
            a.startR[r] = count - 1 + 5;
            count += min(total[r], a.H);                        // columns >= Horizon_SCAN are dropped (:609)
            a.endR[r] = count - 1 - 5;
        }
        a.ringBase[a.N_SCAN] = count;
        *a.d_n = count;
    }
}

__global__ __launch_bounds__(256) void org_scatter_kernel(OrgArgs a)
{
    constexpr int NW = 4;
    __shared__ int waveCnt[NW][MAX_N_SCAN];
    if (threadIdx.x < NW * MAX_N_SCAN) (&waveCnt[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int w = wave_id(), l = lane_id();
    const int cbase = blockIdx.x * ORG_TILE + w * (ORG_TILE / NW);
    const uint64_t lt = lanemask_lt();
    constexpr int IT = ORG_TILE / NW / 64;
    lvi_pt p[IT]; float rg[IT]; int ring[IT]; int rk[IT];
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const int i = cbase + j * 64 + l;
        ring[j] = -1; rk[j] = 0;
        if (i < a.n_raw) {
            const lvi_livox_pt q = a.raw[i];
            ring[j] = org_classify(a, q, &rg[j]);
            p[j].x = q.x; p[j].y = q.y; p[j].z = q.z; p[j].intensity = (float)q.reflectivity;      // :254
        }
        const bool valid = ring[j] >= 0;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 5; b++) {
            const bool bit = (ring[j] >> b) & 1;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        if (valid) {
            const int prior = waveCnt[w][ring[j]];
            rk[j] = prior + __popcll(peers & lt);
            if ((peers & lt) == 0) waveCnt[w][ring[j]] = prior + __popcll(peers);
        }
    }
    __syncthreads();
    if (threadIdx.x < a.N_SCAN) {       // wave offsets inside the tile, per ring
        int off = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) { const int c = waveCnt[q][threadIdx.x]; waveCnt[q][threadIdx.x] = off; off += c; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IT; j++) {
        if (ring[j] < 0) continue;
        const int colIdn = a.blockCnt[ring[j] * a.nblk + blockIdx.x] + waveCnt[w][ring[j]] + rk[j];   // columnIdnCountVec (:604-605)
        if (colIdn >= a.H) continue;                                                                   // :609
        const int dst = a.ringBase[ring[j]] + colIdn;
        a.pts[dst] = p[j];
        a.range[dst] = rg[j];
        a.col[dst] = colIdn;
    }
}

// ---------------------------------------------------------------------------------------------
// a-1 + a-2 fused: curvature stencil and the occlusion / parallel-beam marks.  markOccludedPoints
// is a scatter of idempotent set-to-1 writes; it is evaluated here as a gather so that every point
// is written exactly once.
// ---------------------------------------------------------------------------------------------
struct FeatArgs {
    const int* d_n; const float* range; const int* col;
    float* curv; uint8_t *picked, *picked_occl, *surfmask; int8_t* label;
    const int *startR, *endR, *ringBase; const lvi_pt* pts;
    int *sector_idx, *sector_cnt;
    lvi_pt* corner; int* corner_idx; int* d_ncorner;
    int* d_fresh; int* d_status;
    VoxSegDyn* ringDyn; VoxSegDyn* scanDyn; const int* ringNout;
    int N_SCAN; float edgeThreshold, surfThreshold;
};

__device__ __forceinline__ bool occl_A(const float* r, const int* col, int i)     // depth1 - depth2 > 0.3 at loop index i
{
    return abs(col[i + 1] - col[i]) < 10 && (double)sub_rn(r[i], r[i + 1]) > 0.3;
}
__device__ __forceinline__ bool occl_B(const float* r, const int* col, int i)     // else-if branch
{
    return abs(col[i + 1] - col[i]) < 10 && !((double)sub_rn(r[i], r[i + 1]) > 0.3) && (double)sub_rn(r[i + 1], r[i]) > 0.3;
}

__global__ __launch_bounds__(256) void feat_smooth_kernel(FeatArgs a)
{
    const int n = *a.d_n;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float* r = a.range;
    float c = 0.f;
    if (k >= 5 && k < n - 5) {
        // diffRange = r[i-2] + r[i-1] - r[i]*4 + r[i+1] + r[i+2], left to right (featureExtraction.cpp:99-101)
        const float d = add_rn(add_rn(sub_rn(add_rn(r[k - 2], r[k - 1]), mul_rn(r[k], 4.f)), r[k + 1]), r[k + 2]);
        c = mul_rn(d, d);
    }
    a.curv[k] = c;
    const int lo = 5, hi = n - 6;          // markOccludedPoints loop range [5, n-6)
    bool pk = false;
    if (k + 1 >= lo && k + 1 < hi) pk |= occl_A(r, a.col, k + 1);       // i = k+1 marks i-1
    if (k >= lo && k < hi) {
        pk |= occl_A(r, a.col, k);                                       // i = k marks i
        const float diff1 = fabsf(sub_rn(r[k - 1], r[k]));
        const float diff2 = fabsf(sub_rn(r[k + 1], r[k]));
        pk |= ((double)diff1 > 0.1 * (double)r[k]) && ((double)diff2 > 0.1 * (double)r[k]);   // parallel beam (:142-146)
    }
    if (k - 1 >= lo && k - 1 < hi) pk |= occl_B(r, a.col, k - 1);       // i = k-1 marks i+1
    if (k - 2 >= lo && k - 2 < hi) pk |= occl_B(r, a.col, k - 2);       // i = k-2 marks i+2
    a.picked[k] = pk ? 1 : 0;
    a.picked_occl[k] = pk ? 1 : 0;
    a.label[k] = 0;
    a.surfmask[k] = 0;
}

// ---------------------------------------------------------------------------------------------
// a-3.  One workgroup per ring walks its 6 sectors in order (a sector's neighbour marks spill up
// to 5 points into the next sector, so sectors of one ring are sequentially dependent; rings are
// independent).  Per sector, with curvature / flags of the sector (+5 halo) resident in LDS:
//   corners: the reference sorts the sector and walks it from the largest curvature down, taking a
//            point if it is still unpicked, at most 40.  Equivalent, without sorting: 40 rounds of a
//            workgroup-wide arg-max over the still-unpicked points, each followed by the ±5 marks.
//   "surf":  the ascending greedy walk (label -1, ±5 marks) is a priority-ordered maximal
//            independent set; it is resolved exactly by fixed-point rounds (a point is decided
//            once all its earlier-in-order neighbours are decided).
// Ties in curvature: the reference's std::sort order is unspecified; here larger index first for
// corners, smaller index first for the surf walk.
// ---------------------------------------------------------------------------------------------
constexpr int FEAT_THREADS = 512;
enum : uint8_t { ST_N = 0, ST_U = 1, ST_L = 2 };

__global__ __launch_bounds__(FEAT_THREADS) void feat_sector_kernel(FeatArgs a)
{
    __shared__ float s_curv[FEAT_SEG_CAP + 16];
    __shared__ uint8_t s_pick[FEAT_SEG_CAP + 16], s_brk[FEAT_SEG_CAP + 16], s_state[FEAT_SEG_CAP + 16];
    __shared__ unsigned long long s_key[FEAT_SEG_CAP];        // corner candidates: (curvature bits << 32) | local index
    __shared__ unsigned short s_sorted[FEAT_SEG_CAP];         // candidates by descending key
    __shared__ int s_ws[FEAT_THREADS / 64 + 2];
    __shared__ int s_any;

    const int ring = blockIdx.x;
    const int n = *a.d_n;
    const int tid = threadIdx.x;
    const int fresh = *a.d_fresh;
    volatile uint8_t* v_pick = s_pick;
    volatile uint8_t* v_state = s_state;

    for (int sec = 0; sec < 6; sec++) {
        const int sR = a.startR[ring], eR = a.endR[ring];
        const int sp = (sR * (6 - sec) + eR * sec) / 6;
        const int ep = (sR * (5 - sec) + eR * (sec + 1)) / 6 - 1;
        int* out_idx = a.sector_idx + (ring * 6 + sec) * CORNERS_PER_SECTOR;
        if (sp >= ep) { if (tid == 0) a.sector_cnt[ring * 6 + sec] = 0; continue; }     // :168
        const int k0 = sp - 5;                       // local index j <-> global k = k0 + j
        const int L = ep - sp + 11;
        if (L > FEAT_SEG_CAP) {
            if (tid == 0) { atomicOr(a.d_status, DEV_ERR_SECTOR_TOO_LARGE); a.sector_cnt[ring * 6 + sec] = 0; }
            continue;
        }
        // ---- load sector + halo
        for (int j = tid; j < L; j += FEAT_THREADS) {
            const int k = k0 + j;
            const bool in = (k >= 0 && k < n);
            s_curv[j] = in ? a.curv[k] : 0.f;
            s_pick[j] = in ? a.picked[k] : 1;
            // brk[j]: column jump between k-1 and k (featureExtraction.cpp:190-191,197-198)
            s_brk[j] = (k >= 1 && k < n) ? (uint8_t)(abs(a.col[k] - a.col[k - 1]) > 10) : 1;
        }
        __syncthreads();
        // slot 4 of the whole cloud is the never-rewritten cloudSmoothness entry {0, ind 0}: it is not a
        // candidate in either walk; k = 4 can only be sp of ring 0, sector 0.
        const int jlo = (sp == 4 && ring == 0 && sec == 0) ? 6 : 5;      // first candidate local index
        const int jhi = 5 + (ep - sp);                                    // local index of ep
        // ---- corners.  The reference sorts [sp,ep) by curvature and walks ep, then the sorted range from
        // the top, taking a point if it is still unpicked (max 40) and marking its +-5 neighbours.
        // Here: compact the candidates (unpicked, curvature > edgeThreshold), rank-sort them in LDS, and
        // let ONE wavefront do the greedy walk, 64 candidates at a time, conflicts resolved by ballot.
        int ncand = 0;
        for (int c0 = jlo; c0 < jhi; c0 += FEAT_THREADS) {
            const int j = c0 + tid;
            const bool cand = j < jhi && s_pick[j] == 0 && s_curv[j] > a.edgeThreshold;
            int tot;
            const int pos = ncand + block_excl_scan<FEAT_THREADS>(cand ? 1 : 0, s_ws, &tot);
            if (cand) s_key[pos] = ((unsigned long long)__float_as_uint(s_curv[j]) << 32) | (unsigned)j;
            ncand += tot;
        }
        __syncthreads();
        for (int i = tid; i < ncand; i += FEAT_THREADS) {
            const unsigned long long mine = s_key[i];
            int rank = 0;
            for (int q = 0; q < ncand; q++) rank += (s_key[q] > mine) ? 1 : 0;     // keys are unique (index in the low word)
            s_sorted[rank] = (unsigned short)(mine & 0xFFFFu);
        }
        __syncthreads();
        if (tid < 64) {
            const int l = tid;
            int taken = 0;
            // one "take": record, label, mark +-5 unless a column break intervenes
            auto take = [&](int win) {
                if (l == 0) { out_idx[taken] = k0 + win; a.label[k0 + win] = 1; v_pick[win] = 1; }
                if (l >= 1 && l <= 5) {
                    bool ok = true;
                    for (int q = 1; q <= l; q++) ok = ok && !s_brk[win + q];
                    if (ok) v_pick[win + l] = 1;
                } else if (l >= 6 && l <= 10) {
                    const int m = l - 5; bool ok = true;
                    for (int q = 1; q <= m; q++) ok = ok && !s_brk[win - q + 1];
                    if (ok) v_pick[win - m] = 1;
                }
                taken++;
            };
            // position ep is outside the sorted range and is visited first (:171,174)
            if (jhi >= jlo && v_pick[jhi] == 0 && s_curv[jhi] > a.edgeThreshold) take(jhi);
            for (int base = 0; base < ncand && taken < CORNERS_PER_SECTOR; base += 64) {
                const int j = (base + l < ncand) ? (int)s_sorted[base + l] : -1;
                bool alive = j >= 0 && v_pick[j] == 0;
                uint64_t m = __ballot(alive);
                while (m && taken < CORNERS_PER_SECTOR) {
                    const int first = __ffsll((long long)m) - 1;
                    const int win = __shfl(j, first, 64);
                    take(win);
                    alive = alive && l != first && v_pick[j] == 0;
                    m = __ballot(alive);
                }
            }
            if (l == 0) a.sector_cnt[ring * 6 + sec] = taken;
        }
        __syncthreads();
        // ---- first scan of a fresh node: the stale entry {0, ind 0} is the first element of the ascending
        // walk of ring 0 / sector 0; it labels ind 0 and marks picked[1..5] (only 5 is a candidate).
        if (jlo == 6 && fresh && tid == 0) {
            bool ok = true;
            for (int k = 1; k <= 5; k++) ok = ok && (abs(a.col[k] - a.col[k - 1]) <= 10);
            if (ok && a.surfThreshold > 0.f) s_pick[6] = 1;             // local index of k = 5
        }
        __syncthreads();
        // ---- surf walk as a fixed point (ascending curvature, position ep last; a point is labelled iff no
        // earlier-in-order reachable neighbour is labelled).  Decisions are monotone, so in-place updates are safe.
        for (int j = tid; j < L; j += FEAT_THREADS)
            s_state[j] = (j >= jlo && j <= jhi && s_pick[j] == 0 && s_curv[j] < a.surfThreshold) ? ST_U : ST_N;
        __syncthreads();
        for (int it = 0; it < FEAT_SEG_CAP; it++) {
            if (tid == 0) s_any = 0;
            __syncthreads();
            bool mine = false;
            for (int j = jlo + tid; j <= jhi; j += FEAT_THREADS) {
                if (v_state[j] != ST_U) continue;
                const float cj = s_curv[j];
                bool blocked = false, wait = false;
                for (int dir = 0; dir < 2; dir++) {
                    for (int q = 1; q <= 5; q++) {
                        const int x = dir ? j + q : j - q;
                        if (dir ? s_brk[j + q] : s_brk[j - q + 1]) break;
                        if (x < jlo || x > jhi) continue;
                        const uint8_t sx = v_state[x];
                        if (sx == ST_N) continue;
                        const float cx = s_curv[x];
                        const bool earlier = (x == jhi) ? false : (j == jhi) ? true : (cx < cj || (cx == cj && x < j));
                        if (!earlier) continue;
                        if (sx == ST_L) blocked = true; else wait = true;
                    }
                }
                if (blocked) v_state[j] = ST_N;
                else if (!wait) v_state[j] = ST_L;
                else mine = true;
            }
            if (mine) s_any = 1;
            __syncthreads();
            const int any = s_any;
            __syncthreads();
            if (!any) break;
        }
        // ---- apply labels and their +-5 marks, write back
        for (int j = jlo + tid; j <= jhi; j += FEAT_THREADS) {
            if (s_state[j] != ST_L) continue;
            a.label[k0 + j] = -1;
            v_pick[j] = 1;
            for (int q = 1; q <= 5; q++) { if (s_brk[j + q]) break; v_pick[j + q] = 1; }
            for (int q = 1; q <= 5; q++) { if (s_brk[j - q + 1]) break; v_pick[j - q] = 1; }
        }
        __syncthreads();
        for (int j = tid; j < L; j += FEAT_THREADS) {
            const int k = k0 + j;
            if (k < 0 || k >= n) continue;
            a.picked[k] = s_pick[j];
            if (j >= 5 && j <= jhi) a.surfmask[k] = (a.label[k] <= 0) ? 1 : 0;       // :231-236 (label[k] <= 0)
        }
        __threadfence_block();
        __syncthreads();
    }
    if (ring == 0 && tid == 0 && n > 16) *a.d_fresh = 0;
}

// corners in output order (ring, sector, pick order) + segment descriptors of the per-ring VoxelGrid
__global__ void feat_finalize_kernel(FeatArgs a)
{
    __shared__ int off[MAX_N_SCAN * 6 + 1];
    const int ns = a.N_SCAN * 6;
    if (threadIdx.x == 0) {
        int o = 0;
        for (int s = 0; s < ns; s++) { off[s] = o; o += a.sector_cnt[s]; }
        off[ns] = o;
        *a.d_ncorner = o;
        for (int r = 0; r < a.N_SCAN; r++) { a.ringDyn[r].in_off = a.ringBase[r]; a.ringDyn[r].n = a.ringBase[r + 1] - a.ringBase[r]; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ns * CORNERS_PER_SECTOR; t += blockDim.x) {
        const int s = t / CORNERS_PER_SECTOR, q = t % CORNERS_PER_SECTOR;
        if (q < a.sector_cnt[s]) {
            const int k = a.sector_idx[s * CORNERS_PER_SECTOR + q];
            a.corner_idx[off[s] + q] = k;
            a.corner[off[s] + q] = a.pts[k];
        }
    }
}

__global__ void scan_ds_prep_kernel(VoxSegDyn* dyn, const int* d_ncorner, const int* d_nsurf)
{
    dyn[0].in_off = 0; dyn[0].n = *d_ncorner;
    dyn[1].in_off = 0; dyn[1].n = *d_nsurf;
}

template <class AR>
void layout(AR& ar, LidarDev& d)
{
    const int NS = d.P.N_SCAN;
    d.raw = ar.template alloc<lvi_livox_pt>(d.raw_cap);
    d.blockCnt = ar.template alloc<int>((size_t)NS * d.nblk_org);
    d.ringBase = ar.template alloc<int>(NS + 1);
    d.startR = ar.template alloc<int>(NS); d.endR = ar.template alloc<int>(NS); d.d_n = ar.template alloc<int>(1);
    d.pts = ar.template alloc<lvi_pt>(d.ext_cap); d.range = ar.template alloc<float>(d.ext_cap); d.col = ar.template alloc<int>(d.ext_cap);
    d.curv = ar.template alloc<float>(d.ext_cap);
    d.picked = ar.template alloc<uint8_t>(d.ext_cap); d.picked_occl = ar.template alloc<uint8_t>(d.ext_cap);
    d.surfmask = ar.template alloc<uint8_t>(d.ext_cap); d.label = ar.template alloc<int8_t>(d.ext_cap);
    d.sector_idx = ar.template alloc<int>((size_t)NS * 6 * CORNERS_PER_SECTOR); d.sector_cnt = ar.template alloc<int>(NS * 6);
    d.corner = ar.template alloc<lvi_pt>(d.ext_cap); d.corner_idx = ar.template alloc<int>((size_t)NS * 6 * CORNERS_PER_SECTOR);
    d.d_ncorner = ar.template alloc<int>(1);
    d.surf = ar.template alloc<lvi_pt>(d.ext_cap);
    d.d_fresh = ar.template alloc<int>(1); d.d_status = ar.template alloc<int>(1);
    d.voxRing.allocate(ar, NS, d.ring_cap, true);
    d.cornerDS = ar.template alloc<lvi_pt>(d.ext_cap); d.surfDS = ar.template alloc<lvi_pt>(d.ext_cap);
    d.voxScan.allocate(ar, 2, d.ext_cap, false);
    d.mapCornerRaw = ar.template alloc<lvi_pt>(d.map_cap); d.mapSurfRaw = ar.template alloc<lvi_pt>(d.map_cap);
    d.mapCornerDS = ar.template alloc<lvi_pt>(d.map_cap); d.mapSurfDS = ar.template alloc<lvi_pt>(d.map_cap);
    d.voxMap.allocate(ar, 2, d.map_cap, false);
    for (int w = 0; w < 2; w++) {
        d.grid[w].cell_start = ar.template alloc<int>((size_t)d.max_cells + 2);
        d.grid[w].sorted = ar.template alloc<lvi_pt>(d.map_cap);
        d.grid[w].meta = ar.template alloc<GridIndex::Meta>(1);
    }
    d.gridSort.allocate(ar, 2, d.map_cap);
    d.d_grid_n = ar.template alloc<int>(2); d.d_grid_nbits = ar.template alloc<int>(2);
    const int gen_cap = std::max(d.raw_cap, d.map_cap);
    d.genIn = ar.template alloc<lvi_pt>(gen_cap); d.genOut = ar.template alloc<lvi_pt>(gen_cap);
    d.voxGen.allocate(ar, 1, gen_cap, false);
    d.genKeysDbg = ar.template alloc<unsigned>(gen_cap);
    d.icp = ar.template alloc<IcpState>(1);
    d.icpPartial = ar.template alloc<double>((size_t)d.nblk_icp * 28);
    d.coeff = ar.template alloc<lvi_pt>(d.ext_cap); d.flag = ar.template alloc<uint8_t>(d.ext_cap);
}

FeatArgs feat_args(LidarDev& d)
{
    FeatArgs a{};
    a.d_n = d.d_n; a.range = d.range; a.col = d.col;
    a.curv = d.curv; a.picked = d.picked; a.picked_occl = d.picked_occl; a.surfmask = d.surfmask; a.label = d.label;
    a.startR = d.startR; a.endR = d.endR; a.ringBase = d.ringBase; a.pts = d.pts;
    a.sector_idx = d.sector_idx; a.sector_cnt = d.sector_cnt;
    a.corner = d.corner; a.corner_idx = d.corner_idx; a.d_ncorner = d.d_ncorner;
    a.d_fresh = d.d_fresh; a.d_status = d.d_status;
    a.ringDyn = d.voxRing.d_dyn; a.scanDyn = d.voxScan.d_dyn; a.ringNout = d.voxRing.d_nout;
    a.N_SCAN = d.P.N_SCAN; a.edgeThreshold = d.P.edgeThreshold; a.surfThreshold = d.P.surfThreshold;
    return a;
}

}  // namespace

void lidar_allocate(LidarDev& d)
{
    d.raw_cap = std::max(d.P.max_raw_points, 64);
    d.ring_cap = d.P.Horizon_SCAN;
    const long long full = (long long)d.P.N_SCAN * d.P.Horizon_SCAN;
    d.ext_cap = (int)std::max<long long>(std::min<long long>(d.raw_cap, full), 64);
    d.map_cap = std::max(d.P.max_map_points, 64);
    d.nblk_org = div_up(d.raw_cap, ORG_TILE);
    d.max_cells = 1 << 24;
    d.nblk_icp = div_up(d.ext_cap, ICP_BLOCK / 8);      // 8 lanes per query (KNN_G)
    ArenaSizer sz;
    layout(sz, d);
    d.arena.init(sz.used + (1 << 20));
    layout(d.arena, d);
    LVI_HIP(hipMemsetAsync(d.arena.base, 0, d.arena.size, d.ctx.stream));
    const int one = 1;
    LVI_HIP(hipMemcpyAsync(d.d_fresh, &one, sizeof(int), hipMemcpyHostToDevice, d.ctx.stream));
    LVI_HIP(hipHostMalloc((void**)&d.h_icp, sizeof(IcpState), hipHostMallocDefault));
    // static segment tables of the voxel plans
    std::vector<VoxSegStatic> st(std::max(d.P.N_SCAN, 2));
    for (int r = 0; r < d.P.N_SCAN; r++) st[r] = VoxSegStatic{d.pts, d.surfmask, d.surf, d.P.odometrySurfLeafSize};
    d.voxRing.set_static(d.ctx, st.data());
    st[0] = VoxSegStatic{d.corner, nullptr, d.cornerDS, d.P.mappingCornerLeafSize};
    st[1] = VoxSegStatic{d.surf, nullptr, d.surfDS, d.P.mappingSurfLeafSize};
    d.voxScan.set_static(d.ctx, st.data());
    st[0] = VoxSegStatic{d.mapCornerRaw, nullptr, d.mapCornerDS, d.P.mappingCornerLeafSize};
    st[1] = VoxSegStatic{d.mapSurfRaw, nullptr, d.mapSurfDS, d.P.mappingSurfLeafSize};
    d.voxMap.set_static(d.ctx, st.data());
    LVI_HIP(hipStreamSynchronize(d.ctx.stream));
}

void stage_organize(LidarDev& d)
{
    OrgArgs a{d.raw, d.n_raw, d.blockCnt, d.nblk_org, d.ringBase, d.startR, d.endR, d.d_n, d.pts, d.range, d.col,
              d.P.N_SCAN, d.P.Horizon_SCAN, d.P.downsampleRate, d.P.lidarMinRange, d.P.lidarMaxRange, d.d_status};
    const int nb = std::max(1, div_up(d.n_raw, ORG_TILE));
    const double n = d.n_raw;
    LVI_LAUNCH(d.ctx, "org_count", 20.0 * n, hipLaunchKernelGGL(org_count_kernel, dim3(nb), dim3(256), 0, d.ctx.stream, a));
    LVI_LAUNCH(d.ctx, "org_scan", 0, hipLaunchKernelGGL(org_scan_kernel, dim3(1), dim3(256), 0, d.ctx.stream, a));
    LVI_LAUNCH(d.ctx, "org_scatter", 20.0 * n + 24.0 * n, hipLaunchKernelGGL(org_scatter_kernel, dim3(nb), dim3(256), 0, d.ctx.stream, a));
}

void stage_extract(LidarDev& d)
{
    FeatArgs a = feat_args(d);
    const double n = d.n_raw;
    LVI_LAUNCH(d.ctx, "feat_smooth", 8.0 * n + 8.0 * n, hipLaunchKernelGGL(feat_smooth_kernel, dim3(div_up(d.ext_cap, 256)), dim3(256), 0, d.ctx.stream, a));
    LVI_LAUNCH(d.ctx, "feat_sector", 8.0 * n, hipLaunchKernelGGL(feat_sector_kernel, dim3(d.P.N_SCAN), dim3(FEAT_THREADS), 0, d.ctx.stream, a));
    LVI_LAUNCH(d.ctx, "feat_finalize", 0, hipLaunchKernelGGL(feat_finalize_kernel, dim3(1), dim3(256), 0, d.ctx.stream, a));
    voxel_downsample_batch(d.ctx, d.voxRing, "ring", n);
}

void stage_downsample(LidarDev& d)
{
    LVI_LAUNCH(d.ctx, "scan_ds_prep", 0, hipLaunchKernelGGL(scan_ds_prep_kernel, dim3(1), dim3(1), 0, d.ctx.stream,
                                                           d.voxScan.d_dyn, d.d_ncorner, d.voxRing.d_nout + d.P.N_SCAN));
    voxel_downsample_batch(d.ctx, d.voxScan, "scan", 0.4 * d.n_raw);
}

}  // namespace lvi
