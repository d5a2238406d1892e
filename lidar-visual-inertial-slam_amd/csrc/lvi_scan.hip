// Scan-side kernels of the lidar path for gfx950 (SURVEY §8 a-0 … a-4):
//   organise   ImageProjection::projectPointCloud + cloudExtraction  (imageProjection.cpp:570-647)
//   smooth     FeatureExtraction::calculateSmoothness + markOccludedPoints (featureExtraction.cpp:87-148)
//   sectors    FeatureExtraction::extractFeatures greedy selection   (featureExtraction.cpp:158-237)
//   per-ring / per-scan VoxelGrid                                     (:239-243, mapOptimization.cpp:987-999)
// All counts stay in device memory; the host only enqueues.
#include <cstdlib>

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
    // f-1
    int dk_on, dk_cur; double dk_t0; const double* dk; int* dk_first; float* dk_startInv;
};

// ---------------------------------------------------------------------------------------------
// f-1.  deskewPoint (imageProjection.cpp:538-568): rotation of the point's time stamp from the IMU table
// (findRotation :495-520: the serial scan for the first imuTime > pointTime is a binary search, the table is
// ascending), Rt = getTransformation(0,0,0,rot), point' = (R0^-1 * Rt) * point with R0 of the first point that
// reaches deskewPoint.  findPosition returns zeros in the reference, so the translations are zero.
// sin/cos are taken in double and rounded once: the correctly rounded f32 value the CPU libm returns.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dk_find_rotation(const OrgArgs& a, double pointTime, float rot[3])
{
    const double* T = a.dk;
    int lo = 0, hi = a.dk_cur;                       // first index in [0, cur) whose imuTime > pointTime, else cur
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (pointTime < T[mid]) hi = mid; else lo = mid + 1; }
    const int f = lo;
    if (pointTime > T[f] || f == 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) rot[d] = (float)a.dk[(1 + d) * LVI_DESKEW_MAX_IMU + f];
    } else {
        const int b = f - 1;
        const double ratioFront = (pointTime - T[b]) / (T[f] - T[b]);
        const double ratioBack = (T[f] - pointTime) / (T[f] - T[b]);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double* R = a.dk + (1 + d) * LVI_DESKEW_MAX_IMU;
            rot[d] = (float)(R[f] * ratioFront + R[b] * ratioBack);
        }
    }
}
__device__ __forceinline__ void dk_rotation(const float rot[3], float R[3][3])      // pcl::getTransformation(0,0,0,roll,pitch,yaw), linear part
{
    const float roll = rot[0], pitch = rot[1], yaw = rot[2];
    const float A = (float)cos((double)yaw), B = (float)sin((double)yaw), C = (float)cos((double)pitch), D = (float)sin((double)pitch),
                E = (float)cos((double)roll), F = (float)sin((double)roll), DE = D * E, DF = D * F;
    R[0][0] = A * C; R[0][1] = A * DF - B * E; R[0][2] = B * F + A * DE;
    R[1][0] = B * C; R[1][1] = A * E + B * DF; R[1][2] = B * DE - A * F;
    R[2][0] = -D;    R[2][1] = C * F;          R[2][2] = C * E;
}
__device__ __forceinline__ double dk_point_time(const OrgArgs& a, unsigned offset_time)
{
    const float relTimeF = (float)((double)offset_time * 1e-9);          // PointXYZIRT::time is a float (:255)
    return a.dk_t0 + (double)relTimeF;
}
#define DK_COF(M, i, j) (M[((i) + 1) % 3][((j) + 1) % 3] * M[((i) + 2) % 3][((j) + 2) % 3] - M[((i) + 1) % 3][((j) + 2) % 3] * M[((i) + 2) % 3][((j) + 1) % 3])

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

__global__ __launch_bounds__(256) void org_count_kernel(Batch<OrgArgs> B_)
{
    const OrgArgs& a = B_.a[blockIdx.z];
    __shared__ int cnt[MAX_N_SCAN];
    if (threadIdx.x < MAX_N_SCAN) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * ORG_TILE;
    int first = 0x7fffffff;
    lvi_livox_pt q[ORG_TILE / 256];                     // unconditional loads (clamped index): the tile's points in flight together
    if (a.n_raw > 0) {                                  // (an empty scan still writes its zero counts below)
#pragma unroll
        for (int j = 0; j < ORG_TILE / 256; j++) q[j] = a.raw[min(base + j * 256 + (int)threadIdx.x, a.n_raw - 1)];
    }
#pragma unroll
    for (int j = 0; j < ORG_TILE / 256; j++) {
        const int i = base + j * 256 + threadIdx.x;
        if (i < a.n_raw) {
            float r;
            const int ring = org_classify(a, q[j], &r);
            if (ring >= 0) { atomicAdd(&cnt[ring], 1); first = min(first, i); }
        }
    }
    __syncthreads();
    if (threadIdx.x < a.N_SCAN) a.blockCnt[threadIdx.x * a.nblk + blockIdx.x] = cnt[threadIdx.x];
    if (a.dk_on) {
        // firstPointFlag (:554): the first point in message order that passes the gates (its column is 0 < Horizon_SCAN);
        // one atomic per workgroup (same-address atomics serialise)
        __shared__ int sfirst[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        if (lane_id() == 0) sfirst[wave_id()] = first;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int f = min(min(sfirst[0], sfirst[1]), min(sfirst[2], sfirst[3]));
            if (f != 0x7fffffff) atomicMin(a.dk_first, f);
        }
    }
}

__global__ __launch_bounds__(256) void org_scan_kernel(Batch<OrgArgs> B_)
{
    const OrgArgs& a = B_.a[blockIdx.z];
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
        if (a.dk_on) {
            const int fi = *a.dk_first;
            *a.dk_first = 0x7fffffff;                           // ready for the next scan
            float Ri[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
            if (fi >= 0 && fi < a.n_raw) {
                float rot[3], R[3][3];
                dk_find_rotation(a, dk_point_time(a, a.raw[fi].offset_time), rot);
                dk_rotation(rot, R);
                // Eigen::Affine3f::inverse(): cofactor inverse of the 3x3 linear part (Inverse.h, compute_inverse_size3)
                const float c0 = DK_COF(R, 0, 0), c1 = DK_COF(R, 1, 0), c2 = DK_COF(R, 2, 0);
                const float det = (c0 * R[0][0] + c1 * R[1][0]) + c2 * R[2][0];
                const float invdet = 1.0f / det;
                Ri[0][0] = c0 * invdet; Ri[0][1] = c1 * invdet; Ri[0][2] = c2 * invdet;
                Ri[1][0] = DK_COF(R, 0, 1) * invdet; Ri[1][1] = DK_COF(R, 1, 1) * invdet; Ri[1][2] = DK_COF(R, 2, 1) * invdet;
                Ri[2][0] = DK_COF(R, 0, 2) * invdet; Ri[2][1] = DK_COF(R, 1, 2) * invdet; Ri[2][2] = DK_COF(R, 2, 2) * invdet;
            }
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a.dk_startInv[i * 3 + j] = Ri[i][j];
        }
    }
}

__global__ __launch_bounds__(256) void org_scatter_kernel(Batch<OrgArgs> B_)
{
    const OrgArgs& a = B_.a[blockIdx.z];
    constexpr int NW = 4;
    __shared__ int waveCnt[NW][MAX_N_SCAN];
    if (threadIdx.x < NW * MAX_N_SCAN) (&waveCnt[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int w = wave_id(), l = lane_id();
    const int cbase = blockIdx.x * ORG_TILE + w * (ORG_TILE / NW);
    const uint64_t lt = lanemask_lt();
    constexpr int IT = ORG_TILE / NW / 64;
    lvi_pt p[IT]; float rg[IT]; int ring[IT]; int rk[IT];
    if (a.n_raw <= 0) return;
    lvi_livox_pt qs[IT];                                // unconditional loads (clamped index): the wavefront's points in flight together
#pragma unroll
    for (int j = 0; j < IT; j++) qs[j] = a.raw[min(cbase + j * 64 + l, a.n_raw - 1)];
    // destination base of every ring for this tile (columnIdnCountVec before the tile + the ring's start): one lookup per point
    __shared__ int ringDst[MAX_N_SCAN];
    __shared__ int ringBaseS[MAX_N_SCAN];
    if (threadIdx.x < a.N_SCAN) { ringDst[threadIdx.x] = a.blockCnt[threadIdx.x * a.nblk + blockIdx.x]; ringBaseS[threadIdx.x] = a.ringBase[threadIdx.x]; }
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const int i = cbase + j * 64 + l;
        ring[j] = -1; rk[j] = 0;
        if (i < a.n_raw) {
            const lvi_livox_pt q = qs[j];
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
        const int colIdn = ringDst[ring[j]] + waveCnt[w][ring[j]] + rk[j];                             // columnIdnCountVec (:604-605)
        if (colIdn >= a.H) continue;                                                                   // :609
        const int dst = ringBaseS[ring[j]] + colIdn;
        if (a.dk_on) {
            const int i = cbase + j * 64 + l;
            float rot[3], R[3][3], M[3][3];
            dk_find_rotation(a, dk_point_time(a, a.raw[i].offset_time), rot);
            dk_rotation(rot, R);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++)           // transBt = transStartInverse * transFinal (:561), three products summed left to right
                    M[r][c] = (a.dk_startInv[r * 3] * R[0][c] + a.dk_startInv[r * 3 + 1] * R[1][c]) + a.dk_startInv[r * 3 + 2] * R[2][c];
            const lvi_pt q = p[j];
            p[j].x = M[0][0] * q.x + M[0][1] * q.y + M[0][2] * q.z + 0.f;       // :564-566
            p[j].y = M[1][0] * q.x + M[1][1] * q.y + M[1][2] * q.z + 0.f;
            p[j].z = M[2][0] * q.x + M[2][1] * q.y + M[2][2] * q.z + 0.f;
        }
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
    unsigned* spill;            // [N_SCAN*6][2] sector hand-over words (see feat_sector_kernel), zeroed by feat_smooth
    uint8_t* pflags;            // bit 0: occlusion mark (= initial picked), bit 1: column jump between k-1 and k; static input of the sector kernel
    const int *startR, *endR, *ringBase; const lvi_pt* pts;
    int *sector_idx, *sector_cnt;
    lvi_pt* corner; int* corner_idx; int* d_ncorner;
    int* d_fresh; int* d_status; long long* cyc;
    VoxSegDyn* ringDyn; VoxSegDyn* scanDyn; const int* ringNout;
    int N_SCAN; float edgeThreshold, surfThreshold;
    long long handover_ticks;   // wall_clock64 ticks (100 MHz) a pipelined sector workgroup waits for its predecessor before it redoes the ring itself
};

__device__ __forceinline__ bool occl_A(const float* r, const int* col, int i)     // depth1 - depth2 > 0.3 at loop index i
{
    return abs(col[i + 1] - col[i]) < 10 && (double)sub_rn(r[i], r[i + 1]) > 0.3;
}
__device__ __forceinline__ bool occl_B(const float* r, const int* col, int i)     // else-if branch
{
    return abs(col[i + 1] - col[i]) < 10 && !((double)sub_rn(r[i], r[i + 1]) > 0.3) && (double)sub_rn(r[i + 1], r[i]) > 0.3;
}

__global__ __launch_bounds__(256) void feat_smooth_kernel(Batch<FeatArgs> B_)
{
    const FeatArgs& a = B_.a[blockIdx.z];
    const int n = *a.d_n;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) for (int q = threadIdx.x; q < a.N_SCAN * 12; q += 256) a.spill[q] = 0u;
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
    // column jump between k-1 and k (featureExtraction.cpp:190-191,197-198); the cloud edge counts as one
    const bool brk = k >= 1 ? (abs(a.col[k] - a.col[k - 1]) > 10) : true;
    a.pflags[k] = (uint8_t)((pk ? 1 : 0) | (brk ? 2 : 0));
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
constexpr int FEAT_EPT = FEAT_SEG_CAP / FEAT_THREADS;      // elements per thread (16)
static_assert(FEAT_EPT == 16, "the fixed-point phase assumes 16 contiguous elements (one bitmap halfword) per thread");

// bits [pos, pos+64) of a bitmap stored as 64-bit words (pos >= 0; one word of padding behind the data)
__device__ __forceinline__ uint64_t bits_from(const uint64_t* words, int pos)
{
    const int wi = pos >> 6, sh = pos & 63;
    const uint64_t lo = words[wi];
    return sh ? ((lo >> sh) | (words[wi + 1] << (64 - sh))) : lo;
}

#define CV(j) s_curv[(j) + ((j) >> 4)]
__global__ __launch_bounds__(FEAT_THREADS) void feat_sector_kernel(Batch<FeatArgs> B_)
{
    const FeatArgs& a = B_.a[blockIdx.z];
    __shared__ float s_curv[FEAT_SEG_CAP + 32 + (FEAT_SEG_CAP + 32) / 16 + 2];      // padded: one word behind every 16 (CV below) — a thread's 16 contiguous values
                                                                                     // no longer sit in two banks for the whole wavefront (the static part of a sector was 32-way conflicts)
    __shared__ __attribute__((aligned(16))) uint8_t s_pick[FEAT_SEG_CAP + 16];     // (16-byte aligned: a thread reads its chunk of 16 in one access)
    __shared__ __attribute__((aligned(16))) uint8_t s_reach[FEAT_SEG_CAP + 16];
    __shared__ int8_t s_label[FEAT_SEG_CAP + 16];
    __shared__ uint64_t s_brk[FEAT_SEG_CAP / 64 + 2];         // bit j: column jump (or cloud edge) between j-1 and j
    __shared__ unsigned s_UL[FEAT_THREADS + 4];               // per thread chunk of 16 points: undecided bits (low half) and labelled bits (high half) in ONE word,
                                                              // so that a neighbour reads a consistent pair (+1 pad in front)
    __shared__ unsigned short s_cand[FEAT_SEG_CAP];           // corner candidates (local indices); their order key is (curvature bits, index)
    __shared__ unsigned short s_sorted[FEAT_SEG_CAP];         // candidates by descending key
    __shared__ unsigned short s_cs[FEAT_SEG_CAP + 16];        // corner walk: the candidates of the 64-lane batch in flight, by position (zero outside a batch)
    __shared__ int s_ws[FEAT_THREADS / 64 + 2];
    __shared__ int s_anyr[3], s_timeout;

    // One workgroup per (ring, sector) when every sector of the ring is a regular one; the six workgroups of a ring
    // form a pipeline: loading, neighbour reach, candidate compaction and ranking of sector s+1 run while sector s is
    // in its greedy walks; what s+1 needs from s — the marks on its first five points — arrives in ONE device-scope
    // atomic word (5 flag bits + a ready bit), so no memory fence (= L2 write-back on this multi-XCD part) is involved.
    // HIP does not promise that a producer workgroup is resident before its consumer (several such launches in flight
    // from different streams can fill every CU with waiting consumers), so no workgroup DEPENDS on another one: a consumer
    // waits a bounded wall-clock time (FEAT_HANDOVER_TICKS) for the word and then walks the ring from sector 0 itself
    // ("redo": the same deterministic computation its predecessors do, carry through LDS, identical values written),
    // which always terminates.  Rings with a degenerate sector (< 66 points) take the sequential form from the start:
    // the sector-0 workgroup walks all six.
    const int ring = blockIdx.x / 6, my_sec = blockIdx.x % 6;
    const int n = *a.d_n;
    const int tid = threadIdx.x;
    const int fresh = *a.d_fresh;
    volatile uint8_t* v_pick = s_pick;
    const bool stamp = (blockIdx.x == 0 && tid == 0);
    long long t_prev = stamp ? clock64() : 0, cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define LVI_STAMP(slot) do { if (stamp) { const long long t_now = clock64(); cyc[slot] += t_now - t_prev; t_prev = t_now; } } while (0)

    if (tid < 4) s_UL[tid == 0 ? 0 : FEAT_THREADS + tid] = 0;
    if (tid == 0) { s_brk[FEAT_SEG_CAP / 64] = ~0ull, s_brk[FEAT_SEG_CAP / 64 + 1] = ~0ull; s_timeout = 0; }
    for (int j = FEAT_SEG_CAP + tid; j < FEAT_SEG_CAP + 32; j += FEAT_THREADS) CV(j) = 0.f;
    for (int j = tid; j < FEAT_SEG_CAP + 16; j += FEAT_THREADS) s_cs[j] = 0;
    __syncthreads();

    const int sR = a.startR[ring], eR = a.endR[ring];
    // The global data of sector s+1 (curvature + the flag byte feat_smooth left: 2 loads per point, all static) is
    // fetched into registers while sector s is being processed — measured, the load phase was 20 % of the kernel.
    // The marks earlier sectors add can only reach the 10 points two sectors share; they are carried over from LDS
    // (carry_*).  The raw loaded values stay in registers untouched, so nothing waits for them before the commit.
    float pf_cv[FEAT_EPT]; uint8_t pf_fl[FEAT_EPT];
    auto bounds = [&](int sec, int& sp, int& ep) {
        sp = (sR * (6 - sec) + eR * sec) / 6;
        ep = (sR * (5 - sec) + eR * (sec + 1)) / 6 - 1;
        return sp < ep && ep - sp + 11 <= FEAT_SEG_CAP;
    };
    auto prefetch = [&](int sec) {
        int sp, ep;
        if (!bounds(sec, sp, ep)) return;
        const int k0 = sp - 5, L = ep - sp + 11;
#pragma unroll
        for (int i = 0; i < FEAT_EPT; i++) {        // unconditional loads from a clamped index (behind `in ? … : …` every load waited for the one before it:
            const int j = tid + i * FEAT_THREADS, k = k0 + j;         // 32 round trips at the head of a ring's first sector), lanes masked afterwards
            const int kc = min(max(k, 0), max(n - 1, 0));
            pf_cv[i] = a.curv[kc]; pf_fl[i] = a.pflags[kc];
        }
#pragma unroll
        for (int i = 0; i < FEAT_EPT; i++) {
            const int j = tid + i * FEAT_THREADS, k = k0 + j;
            const bool in = (j < L && k >= 0 && k < n);
            pf_cv[i] = in ? pf_cv[i] : 0.f; pf_fl[i] = in ? pf_fl[i] : (uint8_t)3;
        }
    };
    // pipelined only for up to 4 rings (24 workgroups, each owning a CU's LDS): with many more, several such kernels in flight
    // from different streams could fill every CU with waiting consumers while their producers still wait for a CU
    bool piped = a.N_SCAN * 6 <= 24;
    for (int q = 0; q < 6; q++) { int sp_, ep_; piped = piped && bounds(q, sp_, ep_); }
    // a ring with a sector beyond the LDS-resident capacity belongs to feat_sector_big_kernel (global-memory walk)
    for (int q = 0; q < 6; q++) {
        const int sp_ = (sR * (6 - q) + eR * q) / 6, ep_ = (sR * (5 - q) + eR * (q + 1)) / 6 - 1;
        if (sp_ < ep_ && ep_ - sp_ + 11 > FEAT_SEG_CAP) return;
    }
    if (!piped && my_sec != 0) return;
    const int sec_begin = piped ? my_sec : 0, sec_end = piped ? my_sec + 1 : 6;
    prefetch(sec_begin);
    int carry_k = -1; uint8_t carry_v = 0;            // threads 0..9: picked flags of global points carry_k .. carry_k + 9 as the last sector left them
    bool redo = false;                                // pipelined workgroup that gave up waiting: walks sectors 0 .. my_sec itself

    for (int sec = sec_begin; sec < sec_end; sec++) {
        int sp, ep;
        const bool runnable = bounds(sec, sp, ep);
        int* out_idx = a.sector_idx + (ring * 6 + sec) * CORNERS_PER_SECTOR;
        if (!runnable) {
            if (tid == 0) {
                a.sector_cnt[ring * 6 + sec] = 0;                                   // :168 (sp >= ep)
                if (sp < ep) atomicOr(a.d_status, DEV_ERR_SECTOR_TOO_LARGE);
            }
            if (sec + 1 < sec_end) prefetch(sec + 1);
            continue;
        }
        const int k0 = sp - 5;                       // local index j <-> global k = k0 + j
        const int L = ep - sp + 11;
        // ---- sector + halo from the prefetch registers; the column break flags of 64 consecutive points are one ballot
#pragma unroll
        for (int i = 0; i < FEAT_EPT; i++) {
            const int j = tid + i * FEAT_THREADS;
            CV(j) = pf_cv[i]; s_pick[j] = (uint8_t)(pf_fl[i] & 1u); s_label[j] = 0;
            const uint64_t m = __ballot((pf_fl[i] >> 1) & 1u);
            if (lane_id() == 0) s_brk[j >> 6] = m;
        }
        __syncthreads();
        if (carry_k >= 0 && tid < 10) { const int j = carry_k + tid - k0; if (j >= 0 && j < L) s_pick[j] = carry_v; }
        if (sec + 1 < sec_end) prefetch(sec + 1);    // in flight during everything below
        __syncthreads();
        // reach of the +-5 neighbour marks of a point (they stop at a column break)
#pragma unroll 4
        for (int i = 0; i < FEAT_EPT; i++) {
            const int j = tid + i * FEAT_THREADS;
            const unsigned fw = (unsigned)(bits_from(s_brk, j + 1) & 31u);                 // brk[j+1..j+5]
            const int f = fw ? (__ffs((int)fw) - 1) : 5;
            // backward: brk[j], brk[j-1], …, brk[j-4] must be clear for 1, 2, …, 5 steps
            const unsigned bw = (j >= 4) ? (unsigned)(bits_from(s_brk, j - 4) & 31u) : (unsigned)(((s_brk[0] << (4 - j)) | ((1u << (4 - j)) - 1u)) & 31u);
            const int bk = bw ? (4 - (31 - __clz((int)bw))) : 5;
            s_reach[j] = (uint8_t)(f | (bk << 4));
        }
        __syncthreads();
        LVI_STAMP(0);
        // slot 4 of the whole cloud is the never-rewritten cloudSmoothness entry {0, ind 0}: it is not a
        // candidate in either walk; k = 4 can only be sp of ring 0, sector 0.
        const int jlo = (sp == 4 && ring == 0 && sec == 0) ? 6 : 5;      // first candidate local index
        const int jhi = 5 + (ep - sp);                                    // local index of ep
        // ---- static part of the surf walk (needs curvature and reach only, so it runs before the hand-over wait):
        // thread t owns the 16 CONTIGUOUS points [16t, 16t+16); for each of them the set of reachable neighbours that
        // come EARLIER in the ascending walk is an 11-bit pattern over j-5..j+5 kept in registers.
        const int jb = tid * FEAT_EPT;
        unsigned short pat[FEAT_EPT];
        unsigned statc = 0;                                              // bit i: point jb+i is in [jlo, jhi] with curvature < surfThreshold
        {
            const uint4 rch4 = *reinterpret_cast<const uint4*>(&s_reach[jb]);      // the chunk's 16 reach bytes in one aligned read
            const unsigned rch[4] = {rch4.x, rch4.y, rch4.z, rch4.w};
            float cw[FEAT_EPT + 10];                                     // curvature of j = jb-5 … jb+20
#pragma unroll
            for (int q = 0; q < FEAT_EPT + 10; q++) { const int j = jb - 5 + q; cw[q] = (j >= 0) ? CV(j) : 0.f; }
#pragma unroll
            for (int i = 0; i < FEAT_EPT; i++) {
                const int j = jb + i;
                unsigned p = 0;
                const bool cand = (j >= jlo && j <= jhi && cw[i + 5] < a.surfThreshold);
                if (cand) {
                    const float cj = cw[i + 5];
                    const int rc = (int)((rch[i >> 2] >> (8 * (i & 3))) & 255u);
                    const int f = rc & 15, bk = rc >> 4;
#pragma unroll
                    for (int q = 1; q <= 5; q++) {
                        if (q <= f && j + q <= jhi) {
                            const bool earlier = (j + q == jhi) ? false : (cw[i + 5 + q] < cj);      // ties go to the smaller index (j)
                            if (earlier) p |= 1u << (5 + q);
                        }
                        if (q <= bk && j - q >= jlo) {
                            const bool earlier = (j == jhi) ? true : (cw[i + 5 - q] <= cj);          // ties go to j-q
                            if (earlier) p |= 1u << (5 - q);
                        }
                    }
                    statc |= 1u << i;
                }
                pat[i] = (unsigned short)p;
            }
        }
        // the patterns transposed: em[d + 5] bit i <-> "the neighbour at offset d of point jb + i is reachable and earlier in the walk"
        // (static as well: formed here, before the hand-over wait, not between the two walks)
        unsigned em[11];
#pragma unroll
        for (int d = 0; d < 11; d++) {
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < FEAT_EPT; i++) m |= (((unsigned)pat[i] >> d) & 1u) << i;
            em[d] = m;
        }
        // ---- corners.  The reference sorts [sp,ep) by curvature and walks ep, then the sorted range from
        // the top, taking a point if it is still unpicked (max 40) and marking its +-5 neighbours.
        // Here: compact the candidates (unpicked, curvature > edgeThreshold), rank-sort them in LDS, and
        // let ONE wavefront do the greedy walk, 64 candidates at a time, conflicts resolved by ballot.
        int ncand;
        {
            int mine = 0;
            for (int j = jlo + tid; j < jhi; j += FEAT_THREADS) mine += (s_pick[j] == 0 && CV(j) > a.edgeThreshold) ? 1 : 0;
            int pos = block_excl_scan<FEAT_THREADS>(mine, s_ws, &ncand);
            for (int j = jlo + tid; j < jhi; j += FEAT_THREADS)
                if (s_pick[j] == 0 && CV(j) > a.edgeThreshold) s_cand[pos++] = (unsigned short)j;
        }
        __syncthreads();
        LVI_STAMP(1);
        // rank = candidates with a larger (curvature bits, index) key; the keys are rebuilt from s_curv (every lane reads the
        // same candidate at the same time: two broadcast reads) instead of being kept as 64-bit words — 64 KB of LDS less,
        // so that a sector workgroup leaves room on its CU for the map kernels of the other scans in flight
        for (int i = tid; i < ncand; i += FEAT_THREADS) {
            const int jm = s_cand[i];
            const unsigned long long mine = ((unsigned long long)__float_as_uint(CV(jm)) << 32) | (unsigned)jm;
            int rank = 0;
            for (int q = 0; q < ncand; q++) {
                const int jq = s_cand[q];
                rank += ((((unsigned long long)__float_as_uint(CV(jq)) << 32) | (unsigned)jq) > mine) ? 1 : 0;     // keys are unique (index in the low word)
            }
            s_sorted[rank] = (unsigned short)jm;
        }
        __syncthreads();
        LVI_STAMP(2);
        if (piped && sec > 0 && !redo) {
            // hand-over from sector sec-1: marks on this sector's first five points (local 5..9)
            if (tid == 0) {
                unsigned v = 0u;
                const long long t_give_up = wall_clock64() + a.handover_ticks;
                while (!((v = atomicOr(&a.spill[(ring * 6 + sec - 1) * 2], 0u)) & 0x80000000u)) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() > t_give_up) { s_timeout = 1; break; }
                }
#pragma unroll
                for (int i = 0; i < 5; i++) if ((v >> i) & 1u) s_pick[5 + i] = 1;
            }
            __syncthreads();
            if (s_timeout) {
                // the producer is not making progress (or not resident): do its work here.  Every value written on the way is
                // the one the owner writes (same inputs, same deterministic walk), write-back keeps the pipelined rule.
                redo = true; carry_k = -1;
                prefetch(0);
                sec = -1;
                __syncthreads();
                continue;
            }
        }
        if (tid < 64) {
            const int l = tid;
            int taken = 0;
            // one "take": record, label, mark +-5 unless a column break intervenes
            auto take = [&](int win) {
                const int rc = s_reach[win];
                if (l == 0) { out_idx[taken] = k0 + win; s_label[win] = 1; v_pick[win] = 1; }
                else if (l <= 5) { if (l <= (rc & 15)) v_pick[win + l] = 1; }
                else if (l <= 10) { if (l - 5 <= (rc >> 4)) v_pick[win - (l - 5)] = 1; }
                taken++;
            };
            // position ep is outside the sorted range and is visited first (:171,174)
            if (jhi >= jlo && v_pick[jhi] == 0 && CV(jhi) > a.edgeThreshold) take(jhi);
            // The walk takes the candidates in descending key order, a taken one marks the points within its reach, a marked one is
            // skipped: the lexicographically first maximal independent set of the order.  64 candidates at a time (lane order = walk
            // order), resolved in parallel rounds: a lane is IN once no EARLIER lane that covers it is still undecided or IN, OUT as soon
            // as an earlier IN lane covers it (a later lane's marks come too late to matter to an earlier one).  The lowest undecided
            // lane is always decided, two to four rounds in practice instead of one serial step per take (285 cycles each, 40 per sector:
            // this walk was 40 % of a sector's part of the ring's critical path).  Decisions of earlier lanes never depend on later ones,
            // so cutting the IN set at the 40-corner limit by lane order is the walk stopped at its 40th take.
            for (int base = 0; base < ncand && taken < CORNERS_PER_SECTOR; base += 64) {
                const int j = (base + l < ncand) ? (int)s_sorted[base + l] : -1;
                const int rcj = j >= 0 ? (int)s_reach[j] : 0;
                const bool alive = j >= 0 && s_pick[j] == 0;
                const unsigned short mine = alive ? (unsigned short)((l + 1) | (rcj << 8)) : (unsigned short)0;     // bits 0-6 lane + 1, bit 7 IN, bits 8-15 reach
                if (j >= 0) s_cs[j] = mine;
                bool undec = alive, isin = false;
                for (;;) {
                    __threadfence_block(); __builtin_amdgcn_wave_barrier();      // one wavefront: its LDS operations complete in order
                    bool blocked = false, killed = false;
                    if (undec) {
                        unsigned lo[5], hi[5];
#pragma unroll
                        for (int d = 1; d <= 5; d++) { lo[d - 1] = s_cs[j - d]; hi[d - 1] = s_cs[j + d]; }
#pragma unroll
                        for (int d = 1; d <= 5; d++) {
                            const unsigned p = lo[d - 1], q = hi[d - 1];
                            if (p && (int)(p & 127u) - 1 < l && d <= (int)((p >> 8) & 15u)) { if (p & 128u) killed = true; else blocked = true; }
                            if (q && (int)(q & 127u) - 1 < l && d <= (int)(q >> 12)) { if (q & 128u) killed = true; else blocked = true; }
                        }
                    }
                    const bool now_in = undec && !blocked && !killed, now_out = undec && killed;
                    if (now_in) { isin = true; s_cs[j] = (unsigned short)(mine | 128u); }
                    if (now_out) s_cs[j] = 0;                               // takes nothing, marks nothing, blocks nobody
                    undec = undec && !now_in && !now_out;
                    if (!__ballot(undec)) break;
                }
                const uint64_t inm = __ballot(isin);
                const int room = CORNERS_PER_SECTOR - taken, nin = __popcll(inm);
                const int rank = __popcll(inm & ((1ull << l) - 1ull));
                if (isin && rank < room) {
                    out_idx[taken + rank] = k0 + j; s_label[j] = 1;
                    const int f = rcj & 15, bk = rcj >> 4;
                    for (int q = 0; q <= f; q++) s_pick[j + q] = 1;
                    for (int q = 1; q <= bk; q++) s_pick[j - q] = 1;
                }
                taken += min(nin, room);
                if (j >= 0) s_cs[j] = 0;
                __threadfence_block(); __builtin_amdgcn_wave_barrier();
            }
            if (l == 0) a.sector_cnt[ring * 6 + sec] = taken;
        }
        __syncthreads();
        LVI_STAMP(3);
        // ---- first scan of a fresh node: the stale entry {0, ind 0} is the first element of the ascending
        // walk of ring 0 / sector 0; it labels ind 0 and marks picked[1..5] (only 5 is a candidate).
        if (jlo == 6 && fresh && tid == 0) {
            bool ok = true;
            for (int k = 1; k <= 5; k++) ok = ok && (abs(a.col[k] - a.col[k - 1]) <= 10);
            if (ok && a.surfThreshold > 0.f) s_pick[6] = 1;             // local index of k = 5
            if (a.surfThreshold > 0.f && n > 0) { a.label[0] = -1; a.picked[0] = 1; }      // ind 0 itself (outside every sector: no effect on the outputs)
        }
        __syncthreads();
        // ---- surf walk as a fixed point (ascending curvature, position ep last; a point is labelled iff no
        // earlier-in-order reachable neighbour is labelled).  Thread t owns the 16 CONTIGUOUS points
        // [16t, 16t+16) = one halfword of the "undecided" (U) and "labelled" (L) bitmaps.  For each point the
        // set of reachable neighbours that come earlier in the walk is static: an 11-bit pattern over j-5..j+5
        // kept in registers.  A round reads the three halfwords around the chunk of each bitmap, resolves the
        // chunk with shifts and masks, and stores its own halfwords.  Decisions are monotone, so a neighbour
        // chunk's old or new bits are equally valid.
        unsigned myU = 0, myL = 0;
        {
            const uint4 pk4 = *reinterpret_cast<const uint4*>(&s_pick[jb]);         // the chunk's 16 picked bytes in one aligned read
            const unsigned pk[4] = {pk4.x, pk4.y, pk4.z, pk4.w};
#pragma unroll
            for (int i = 0; i < FEAT_EPT; i++) if (((statc >> i) & 1u) && ((pk[i >> 2] >> (8 * (i & 3))) & 255u) == 0u) myU |= 1u << i;      // candidates the corner walk left unpicked
            s_UL[tid + 1] = myU;
        }
        __syncthreads();
        LVI_STAMP(4);
        int rounds = 0;
        // one barrier per round: three flags in rotation (a round raises flag it % 3, reads it after the barrier, and clears the
        // flag of round it + 2, which nobody touches before the next barrier)
        if (tid < 3) s_anyr[tid] = 0;
        __syncthreads();
        for (int it = 0; it < FEAT_SEG_CAP; it++) {
            if (myU) {
                // 48-bit windows: bit 16 + i <-> point jb + i.  All 16 points of the chunk are resolved together, offset by offset
                // (a point is decided once no earlier reachable neighbour is undecided; labelled iff none of them is labelled);
                // three passes per round let a decision travel inside the chunk before the neighbours see it
                const unsigned wp = s_UL[tid], wn = s_UL[tid + 2];
#pragma unroll
                for (int rep = 0; rep < 3; rep++) {
                    const uint64_t Uw = (uint64_t)(wp & 0xFFFFu) | ((uint64_t)myU << 16) | ((uint64_t)(wn & 0xFFFFu) << 32);
                    const uint64_t Lw = (uint64_t)(wp >> 16) | ((uint64_t)myL << 16) | ((uint64_t)(wn >> 16) << 32);
                    unsigned blocked = 0, wait = 0;
#pragma unroll
                    for (int d = 0; d < 11; d++) {
                        if (d == 5) continue;
                        blocked |= em[d] & (unsigned)(Lw >> (11 + d));
                        wait |= em[d] & (unsigned)(Uw >> (11 + d));
                    }
                    myL |= myU & ~blocked & ~wait;
                    myU &= ~(blocked | ~wait);
                }
                s_UL[tid + 1] = myU | (myL << 16);
                if (myU) s_anyr[it % 3] = 1;
            }
            __syncthreads();
            const int any = s_anyr[it % 3];
            if (tid == 0) s_anyr[(it + 2) % 3] = 0;
            rounds++;
            if (!any) break;
        }
        __syncthreads();
        if (stamp) cyc[6] += rounds;
        LVI_STAMP(5);
        // ---- apply labels and their +-5 marks
        {
            unsigned rem = myL;
            while (rem) {
                const int i = __ffs((int)rem) - 1;
                rem &= rem - 1;
                const int j = jb + i;
                s_label[j] = -1;
                const int rc = s_reach[j];
                v_pick[j] = 1;
                for (int q = 1; q <= (rc & 15); q++) v_pick[j + q] = 1;
                for (int q = 1; q <= (rc >> 4); q++) v_pick[j - q] = 1;
            }
        }
        __syncthreads();
        if (piped && tid == 0) {
            // forward word: marks on the next sector's first five points + ready; backward word: marks this sector made on
            // the previous sector's last five points (feat_finalize ORs them into picked[])
            unsigned fw = 0x80000000u, bw = 0u;
#pragma unroll
            for (int i = 0; i < 5; i++) { if (s_pick[jhi + 1 + i]) fw |= 1u << i; if (s_pick[i]) bw |= 1u << i; }
            if (sec > 0) atomicExch(&a.spill[(ring * 6 + sec) * 2 + 1], bw);
            atomicExch(&a.spill[(ring * 6 + sec) * 2], fw);
        }
        // ---- write back: picked (sequential form: incl. the 5-point spill into the neighbouring sectors; pipelined form:
        // the sector's own points, plus the leading / trailing five of the ring's first / last sector), labels, surf candidates
#pragma unroll 4
        for (int i = 0; i < FEAT_EPT; i++) {
            const int j = tid + i * FEAT_THREADS;
            if (j >= L) break;
            const int k = k0 + j;
            if (k < 0 || k >= n) continue;
            if (piped && ((j < 5 && sec > 0) || (j > jhi && sec < 5))) continue;
            a.picked[k] = s_pick[j];
            if (j >= 5 && j <= jhi) {
                const int8_t lb = s_label[j];
                a.label[k] = lb;
                a.surfmask[k] = (lb <= 0) ? 1 : 0;                       // :231-236 (cloudLabel[k] <= 0)
            }
        }
        carry_k = ep - 4;                            // the 10 points [ep-4, ep+5] are the next sector's halo and first five points
        if (tid < 10) carry_v = s_pick[ep - 4 - k0 + tid];
        __threadfence_block();
        __syncthreads();
        LVI_STAMP(7);
    }
    if (stamp) for (int q = 0; q < 8; q++) a.cyc[q] = cyc[q];
#undef LVI_STAMP
}

// ---------------------------------------------------------------------------------------------
#undef CV
// a-3 for rings whose sectors do not fit the LDS-resident kernel (more than FEAT_SEG_CAP - 11 points per sector, e.g.
// N_SCAN = 1 with > 49 k points): the same two walks over GLOBAL memory, one workgroup per ring, sectors in order (the marks
// of a sector simply stay in picked[] for the next one).  Slow by design — tens of block-wide passes per sector — and exact:
//   corners  up to 40 rounds of "largest (curvature, index) among the still unpicked candidates" = the reference's walk down
//            the sorted range (the ep slot first, :171-174)
//   surf     the same priority-ordered maximal independent set as the LDS kernel, state kept in label[] (2 = undecided)
// Launched only when the handle's geometry allows such a sector (host check), and a no-op for rings the LDS kernel took.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FEAT_THREADS) void feat_sector_big_kernel(Batch<FeatArgs> B_)
{
    const FeatArgs& a = B_.a[blockIdx.z];
    const int ring = blockIdx.x, tid = threadIdx.x;
    const int n = *a.d_n;
    const int sR = a.startR[ring], eR = a.endR[ring];
    bool big = false;
    for (int q = 0; q < 6; q++) {
        const int sp_ = (sR * (6 - q) + eR * q) / 6, ep_ = (sR * (5 - q) + eR * (q + 1)) / 6 - 1;
        big = big || (sp_ < ep_ && ep_ - sp_ + 11 > FEAT_SEG_CAP);
    }
    if (!big) return;
    const int fresh = *a.d_fresh;
    volatile uint8_t* picked = a.picked;
    volatile int8_t* label = a.label;
    const float* __restrict__ curv = a.curv;
    const uint8_t* __restrict__ pfl = a.pflags;                  // bit 1: column break between k-1 and k
    __shared__ unsigned long long s_best[FEAT_THREADS / 64];
    __shared__ int s_flag;
    auto brk = [&](int k) { return k <= 0 || k >= n || ((pfl[k] >> 1) & 1); };      // the cloud's ends count as breaks
    auto fwd_reach = [&](int k) { int f = 0; while (f < 5 && !brk(k + f + 1)) f++; return f; };
    auto bwd_reach = [&](int k) { int b = 0; while (b < 5 && !brk(k - b)) b++; return b; };
    for (int sec = 0; sec < 6; sec++) {
        const int sp = (sR * (6 - sec) + eR * sec) / 6, ep = (sR * (5 - sec) + eR * (sec + 1)) / 6 - 1;
        int* out_idx = a.sector_idx + (ring * 6 + sec) * CORNERS_PER_SECTOR;
        if (sp >= ep) { if (tid == 0) a.sector_cnt[ring * 6 + sec] = 0; continue; }
        // slot 4 of the whole cloud: the never-rewritten cloudSmoothness entry, not a candidate
        const int klo = (sp == 4 && ring == 0 && sec == 0) ? 5 : sp;
        // ---- corners
        int taken = 0;
        for (int round = 0; round < CORNERS_PER_SECTOR + 1; round++) {
            unsigned long long best = 0ull;                       // (curvature bits << 32) | (k - sp + 1); 0 = none
            if (round == 0) {
                if (tid == 0 && picked[ep] == 0 && curv[ep] > a.edgeThreshold) best = ((unsigned long long)__float_as_uint(curv[ep]) << 32) | (unsigned)(ep - sp + 1);
            } else {
                for (int k = klo + tid; k < ep; k += FEAT_THREADS)
                    if (picked[k] == 0 && curv[k] > a.edgeThreshold) {
                        const unsigned long long key = ((unsigned long long)__float_as_uint(curv[k]) << 32) | (unsigned)(k - sp + 1);
                        best = key > best ? key : best;
                    }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(best, o, 64); best = v > best ? v : best; }
            if (lane_id() == 0) s_best[wave_id()] = best;
            __syncthreads();
            best = 0ull;
            for (int w = 0; w < FEAT_THREADS / 64; w++) best = s_best[w] > best ? s_best[w] : best;
            __syncthreads();
            if (best == 0ull) { if (round == 0) continue; break; }
            const int win = sp + (int)(unsigned)(best & 0xFFFFFFFFull) - 1;
            if (tid == 0) {
                out_idx[taken] = win; label[win] = 1; picked[win] = 1;
                const int f = fwd_reach(win), b = bwd_reach(win);
                for (int q = 1; q <= f; q++) picked[win + q] = 1;
                for (int q = 1; q <= b; q++) picked[win - q] = 1;
            }
            taken++;
            __threadfence_block();
            __syncthreads();
            if (taken >= CORNERS_PER_SECTOR) break;
        }
        if (tid == 0) a.sector_cnt[ring * 6 + sec] = taken;
        // ---- first scan of a fresh node (see the LDS kernel)
        if (klo == 5 && fresh && tid == 0) {
            bool ok = true;
            for (int k = 1; k <= 5; k++) ok = ok && (abs(a.col[k] - a.col[k - 1]) <= 10);
            if (ok && a.surfThreshold > 0.f) picked[5] = 1;
            if (a.surfThreshold > 0.f && n > 0) { label[0] = -1; picked[0] = 1; }
        }
        __threadfence_block();
        __syncthreads();
        // ---- surf walk: candidates = unpicked points below the threshold; order = ascending (curvature, index), ep last
        for (int k = klo + tid; k <= ep; k += FEAT_THREADS)
            if (picked[k] == 0 && curv[k] < a.surfThreshold && label[k] == 0) label[k] = 2;
        __threadfence_block();
        __syncthreads();
        auto earlier = [&](int q, int k) {                        // does q come before k in the walk?
            if (q == ep) return false;
            if (k == ep) return true;
            const float cq = curv[q], ck = curv[k];
            return cq < ck || (cq == ck && q < k);
        };
        for (int it = 0; it < n + 2; it++) {
            if (tid == 0) s_flag = 0;
            __syncthreads();
            for (int k = klo + tid; k <= ep; k += FEAT_THREADS) {
                if (label[k] != 2) continue;
                bool blocked = false, wait = false;
                const int f = fwd_reach(k), b = bwd_reach(k);
                for (int q = 1; q <= f && k + q <= ep; q++) {
                    const int8_t lq = label[k + q];
                    if ((lq == -1 || lq == 2) && earlier(k + q, k)) { blocked = blocked || lq == -1; wait = wait || lq == 2; }
                }
                for (int q = 1; q <= b && k - q >= klo; q++) {
                    const int8_t lq = label[k - q];
                    if ((lq == -1 || lq == 2) && earlier(k - q, k)) { blocked = blocked || lq == -1; wait = wait || lq == 2; }
                }
                // decisions are monotone: reading a neighbour's old or new state gives a valid (possibly later) decision
                if (blocked) label[k] = 0;
                else if (!wait) label[k] = -1;
                else s_flag = 1;
            }
            __threadfence_block();
            __syncthreads();
            const int any = s_flag;
            __syncthreads();
            if (!any) break;
        }
        // ---- marks of the labelled points, then the surf candidates of the sector (:231-236)
        for (int k = klo + tid; k <= ep; k += FEAT_THREADS)
            if (label[k] == -1) {
                picked[k] = 1;
                const int f = fwd_reach(k), b = bwd_reach(k);
                for (int q = 1; q <= f; q++) picked[k + q] = 1;
                for (int q = 1; q <= b; q++) picked[k - q] = 1;
            }
        for (int k = sp + tid; k <= ep; k += FEAT_THREADS) a.surfmask[k] = (label[k] <= 0) ? 1 : 0;
        __threadfence_block();
        __syncthreads();
    }
}

// corners in output order (ring, sector, pick order) + segment descriptors of the per-ring VoxelGrid
__global__ void feat_finalize_kernel(Batch<FeatArgs> B_)
{
    const FeatArgs& a = B_.a[blockIdx.z];
    __shared__ int off[MAX_N_SCAN * 6 + 1];
    const int ns = a.N_SCAN * 6;
    __shared__ int scnt[MAX_N_SCAN * 6];
    for (int s = threadIdx.x; s < ns; s += blockDim.x) scnt[s] = a.sector_cnt[s];      // one round trip for all sectors (thread 0 alone: one per sector)
    __syncthreads();
    if (threadIdx.x == 0) {
        int o = 0;
        for (int s = 0; s < ns; s++) { off[s] = o; o += scnt[s]; }
        off[ns] = o;
        *a.d_ncorner = o;
        for (int r = 0; r < a.N_SCAN; r++) { a.ringDyn[r].in_off = a.ringBase[r]; a.ringDyn[r].n = a.ringBase[r + 1] - a.ringBase[r]; }
        if (*a.d_n > 16) *a.d_fresh = 0;              // the "fresh node" case ends with the first real scan
    }
    __syncthreads();
    // pipelined sector kernel: marks a sector made on the last five points of its predecessor
    for (int t = threadIdx.x; t < ns * 5; t += blockDim.x) {
        const int s = t / 5, i = t % 5, ring = s / 6, sec = s % 6;
        if (sec == 0) continue;
        const unsigned bw = a.spill[s * 2 + 1];
        if (!((bw >> i) & 1u)) continue;
        const int sR = a.startR[ring], eR = a.endR[ring];
        const int sp = (sR * (6 - sec) + eR * sec) / 6;
        const int k = sp - 5 + i;
        if (k >= 0 && k < *a.d_n) a.picked[k] = 1;
    }
    for (int t = threadIdx.x; t < ns * CORNERS_PER_SECTOR; t += blockDim.x) {
        const int s = t / CORNERS_PER_SECTOR, q = t % CORNERS_PER_SECTOR;
        if (q < scnt[s]) {
            const int k = a.sector_idx[s * CORNERS_PER_SECTOR + q];
            a.corner_idx[off[s] + q] = k;
            a.corner[off[s] + q] = a.pts[k];
        }
    }
}

template <class AR>
void layout(AR& ar, LidarDev& d)
{
    const int NS = d.P.N_SCAN;
    d.raw = ar.template alloc<lvi_livox_pt>(d.raw_cap);
    d.blockCnt = ar.template alloc<int>((size_t)NS * d.nblk_org);
    d.d_dk = ar.template alloc<double>(4 * LVI_DESKEW_MAX_IMU); d.d_dk_first = ar.template alloc<int>(1); d.d_dk_startInv = ar.template alloc<float>(12);
    d.ringBase = ar.template alloc<int>(NS + 1);
    d.startR = ar.template alloc<int>(NS); d.endR = ar.template alloc<int>(NS); d.d_n = ar.template alloc<int>(1);
    d.pts = ar.template alloc<lvi_pt>(d.ext_cap); d.range = ar.template alloc<float>(d.ext_cap); d.col = ar.template alloc<int>(d.ext_cap);
    d.curv = ar.template alloc<float>(d.ext_cap);
    d.picked = ar.template alloc<uint8_t>(d.ext_cap); d.picked_occl = ar.template alloc<uint8_t>(d.ext_cap); d.pflags = ar.template alloc<uint8_t>(d.ext_cap); d.sectorSpill = ar.template alloc<unsigned>((size_t)MAX_N_SCAN * 12);
    d.surfmask = ar.template alloc<uint8_t>(d.ext_cap); d.label = ar.template alloc<int8_t>(d.ext_cap);
    d.sector_idx = ar.template alloc<int>((size_t)NS * 6 * CORNERS_PER_SECTOR); d.sector_cnt = ar.template alloc<int>(NS * 6);
    d.corner = ar.template alloc<lvi_pt>(d.ext_cap); d.corner_idx = ar.template alloc<int>((size_t)NS * 6 * CORNERS_PER_SECTOR);
    d.d_ncorner = ar.template alloc<int>(1);
    d.surf = ar.template alloc<lvi_pt>(d.ext_cap);
    d.d_fresh = ar.template alloc<int>(1); d.d_status = ar.template alloc<int>(2); d.d_feat_cycles = ar.template alloc<long long>(8); d.d_icp_cycles = ar.template alloc<long long>(16);
    d.voxRing.allocate(ar, NS, d.ring_cap, true);
    d.cornerDS = ar.template alloc<lvi_pt>(d.ext_cap); d.surfDS = ar.template alloc<lvi_pt>(d.ext_cap);
    d.voxScan.allocate(ar, 2, d.ext_cap, false);
    if (d.map_owner) { d.mapCornerRaw = d.map_owner->mapCornerRaw; d.mapSurfRaw = d.map_owner->mapSurfRaw; }
    else { d.mapCornerRaw = d.mapCornerOwn = ar.template alloc<lvi_pt>(d.map_cap); d.mapSurfRaw = d.mapSurfOwn = ar.template alloc<lvi_pt>(d.map_cap); }
    d.mapCornerDS = ar.template alloc<lvi_pt>(d.map_cap); d.mapSurfDS = ar.template alloc<lvi_pt>(d.map_cap);
    d.voxMap.allocate(ar, 2, d.map_cap, false);
    d.voxMap.centroid_lanes = 32;
    if (d.map_owner && d.P.map_plan_cache) {            // … of the shared raw map: one plan for every slot
        d.voxMap.d_mmPartial = d.map_owner->voxMap.d_mmPartial; d.voxMap.d_binCountCached = d.map_owner->voxMap.d_binCountCached;
        d.voxMap.d_wprefix = d.map_owner->voxMap.d_wprefix;
    } else {                                            // every slot takes its own plan inside every re-voxelisation (the default)
        d.voxMap.d_binCountCached = ar.template alloc<unsigned>((size_t)2 * VB_NB);
        d.voxMap.d_wprefix = ar.template alloc<unsigned>((size_t)2 * VB_WG * VB_WROW);
    }
    for (int w = 0; w < 2; w++) {
        d.grid[w].cell_start = ar.template alloc<int>((size_t)d.max_cells + 2);
        d.grid[w].count = ar.template alloc<int>((size_t)d.max_cells + 2);
        d.grid[w].blockSum = ar.template alloc<int>(1024);
        d.grid[w].sorted = ar.template alloc<lvi_pt>(d.map_cap);
        d.grid[w].meta = ar.template alloc<GridIndex::Meta>(1);
    }
    const int gen_cap = d.map_owner ? 64 : std::max(d.raw_cap, d.map_cap);      // one-call voxel / transform entry points run on slot 0
    d.genIn = ar.template alloc<lvi_pt>(gen_cap); d.genOut = ar.template alloc<lvi_pt>(gen_cap);
    d.voxGen.allocate(ar, 1, gen_cap, false);
    d.genKeysDbg = ar.template alloc<unsigned>(gen_cap);
    d.kfPool = ar.template alloc<lvi_pt>((size_t)std::max(d.kf_pool_cap, 1));
    if (d.kf_pool_cap > 0 && d.P.batch_scans <= 1) {
        // incremental local map: slots for 4x the voxels a full map of this capacity typically leaves (tombstones included; a
        // table that fills up is rebuilt), (idx, slot) sort pairs for half of them
        int H = 1 << 16;
        while (H < d.map_cap / 2 && H < (1 << 22)) H <<= 1;
        d.inc.allocate(ar, H, std::max(d.P.max_keyframes, 1), d.kf_seg_cap);
    }
    d.d_kfSeg = ar.template alloc<LidarDev::KfSeg>((size_t)std::max(d.kf_seg_cap, 1));
    d.icp = ar.template alloc<IcpState>(1);
    d.d_pose_init = ar.template alloc<float>(8);
    d.icpAcc = ar.template alloc<unsigned long long>(3 * 8 * 56);
    d.coeff = ar.template alloc<lvi_pt>(d.ext_cap); d.flag = ar.template alloc<uint8_t>(d.ext_cap);
    d.nnPrev = ar.template alloc<int>((size_t)d.ext_cap * 5);
    d.nnRef = ar.template alloc<float4>((size_t)d.ext_cap);
    d.nnPt = ar.template alloc<float4>((size_t)d.ext_cap * 5);
    d.fitA = ar.template alloc<float4>((size_t)d.ext_cap); d.fitB = ar.template alloc<float4>((size_t)d.ext_cap);
    d.fitOk = ar.template alloc<unsigned char>((size_t)d.ext_cap);
}

FeatArgs feat_args(LidarDev& d)
{
    FeatArgs a{};
    a.d_n = d.d_n; a.range = d.range; a.col = d.col;
    a.curv = d.curv; a.picked = d.picked; a.picked_occl = d.picked_occl; a.surfmask = d.surfmask; a.label = d.label; a.pflags = d.pflags; a.spill = d.sectorSpill;
    a.startR = d.startR; a.endR = d.endR; a.ringBase = d.ringBase; a.pts = d.pts;
    a.sector_idx = d.sector_idx; a.sector_cnt = d.sector_cnt;
    a.corner = d.corner; a.corner_idx = d.corner_idx; a.d_ncorner = d.d_ncorner;
    a.d_fresh = d.d_fresh; a.d_status = d.d_status; a.cyc = d.d_feat_cycles;
    a.ringDyn = d.voxRing.d_dyn; a.scanDyn = d.voxScan.d_dyn; a.ringNout = d.voxRing.d_nout;
    a.N_SCAN = d.P.N_SCAN; a.edgeThreshold = d.P.edgeThreshold; a.surfThreshold = d.P.surfThreshold;
    a.handover_ticks = d.feat_handover_ticks;
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
    d.kf_pool_cap = (d.P.max_keyframes > 0 && !d.map_owner) ? std::max(d.P.max_keyframe_points, 0) : 0;
    d.kf_seg_cap = 2 * std::max(d.P.max_keyframes, 0) + 2048;       // an assembly may list a key more than once
    // cells of the KNN grid (0.5 m, or the next multiple that fits): sized from the map capacity — three int arrays per grid
    // and map, 400 MB at 2^24 — so that small handles stay small; a map whose extent needs more gets coarser cells (still exact)
    d.max_cells = (int)std::min<long long>(1ll << 24, std::max<long long>(1ll << 18, 4ll * d.map_cap));
    { const char* e = getenv("LVI_KNN_NO_BOUND"); d.knn_bound = !(e && e[0] == '1'); }
    { const char* e = getenv("LVI_KNN_NO_SKIP"); d.knn_skip = !(e && e[0] == '1'); }
    { const char* e = getenv("LVI_KNN_SLACK"); if (e) d.knn_slack = std::max(0.f, (float)atof(e)); }
    { const char* e = getenv("LVI_VB_BINS"); if (e) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0 && b <= VB_NB) { d.voxMap.bin_pts = a; d.voxMap.bin_max = b; } } }
    if (d.P.batch_scans > 1) { d.icp_g1 = 4; d.icp_wide_from = 1; }      // (256 features per workgroup from iteration 1 on: 8 040 vs 7 930 scans/s; a single scan: 640 vs 612 us)
    // (lanes per feature after iteration 0: round 2's kernel preferred 2 in throughput mode; with the records in global memory both forms
    //  run four wavefronts per SIMD and 4 lanes win: 7 177 vs 7 090 scans/s)
    { const char* e = getenv("LVI_ICP_G0"); if (e) d.icp_g0 = atoi(e); }
    { const char* e = getenv("LVI_KNN_TILES"); d.knn_tiles = e && e[0] == '1'; }
    { const char* e = getenv("LVI_ICP_STAMP_ITER"); if (e) d.icp_stamp_iter = atoi(e); }
    { const char* e = getenv("LVI_ICP_WIDE_FROM"); if (e) d.icp_wide_from = std::max(1, atoi(e)); }
    { const char* e = getenv("LVI_ICP_G1"); if (e) d.icp_g1 = atoi(e); if (!d.knn_bound) d.icp_g1 = 8; }
    d.feat_handover_ticks = d.P.sector_handover_wait_us < 0 ? 0 : 100ll * (d.P.sector_handover_wait_us > 0 ? d.P.sector_handover_wait_us : 2000);
    d.nblk_icp = div_up(d.ext_cap, ICP_BLOCK / KNN_G);
    ArenaSizer sz;
    layout(sz, d);
    d.arena.init(sz.used + (1 << 20));
    layout(d.arena, d);
    LVI_HIP(hipMemsetAsync(d.arena.base, 0, d.arena.size, d.ctx.stream));
    const int one = 1, int_max = 0x7fffffff;
    LVI_HIP(hipMemcpyAsync(d.d_fresh, &one, sizeof(int), hipMemcpyHostToDevice, d.ctx.stream));
    LVI_HIP(hipMemcpyAsync(d.d_dk_first, &int_max, sizeof(int), hipMemcpyHostToDevice, d.ctx.stream));
    LVI_HIP(hipHostMalloc((void**)&d.h_icp, sizeof(IcpState), hipHostMallocDefault));
    LVI_HIP(hipHostMalloc((void**)&d.h_res, sizeof(IcpHostResult), hipHostMallocDefault));
    memset(d.h_res, 0, sizeof(IcpHostResult));
    LVI_HIP(hipHostMalloc((void**)&d.h_gn_feat, 64, hipHostMallocDefault));
    *d.h_gn_feat = 0;
    for (int s = 0; s < 2; s++) {
        LVI_HIP(hipHostMalloc((void**)&d.h_raw[s], sizeof(lvi_livox_pt) * (size_t)d.raw_cap, hipHostMallocDefault));
        LVI_HIP(hipEventCreateWithFlags(&d.ev_raw[s], hipEventDisableTiming));
        LVI_HIP(hipEventRecord(d.ev_raw[s], d.ctx.stream));
    }
    if (d.inc.H) {
        LVI_HIP(hipHostMalloc((void**)&d.inc.h_pieces, sizeof(IncPiece) * (size_t)d.inc.max_pieces, hipHostMallocDefault));
        LVI_HIP(hipHostMalloc((void**)&d.inc.h_status, sizeof(int) * 4, hipHostMallocDefault));
        LVI_HIP(hipHostMalloc((void**)&d.inc.h_active, sizeof(int) * (size_t)std::max(d.inc.max_active, 1), hipHostMallocDefault));
    }
    LVI_HIP(hipHostMalloc((void**)&d.h_kfSeg, sizeof(LidarDev::KfSeg) * (size_t)std::max(d.kf_seg_cap, 1), hipHostMallocDefault));
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
    d.voxRing.mode = d.voxScan.mode = d.voxMap.mode = d.voxGen.mode = d.P.voxel_mode;
    // the scan grids take their input counts straight from the producers' device counters (no 1-thread launch in between)
    d.voxScan.n_dev[0] = d.d_ncorner; d.voxScan.n_dev[1] = d.voxRing.d_nout + d.P.N_SCAN;
    LVI_HIP(hipStreamSynchronize(d.ctx.stream));
}

static OrgArgs org_args(LidarDev& d)
{
    return OrgArgs{d.raw_bound ? d.raw_bound : d.raw, d.n_raw, d.blockCnt, d.nblk_org, d.ringBase, d.startR, d.endR, d.d_n, d.pts, d.range, d.col,
                   d.P.N_SCAN, d.P.Horizon_SCAN, d.P.downsampleRate, d.P.lidarMinRange, d.P.lidarMaxRange, d.d_status,
                   d.dk_on ? 1 : 0, d.dk_cur, d.dk_t0, d.d_dk, d.d_dk_first, d.d_dk_startInv};
}

void stage_organize(const Slots& sl)
{
    Batch<OrgArgs> B;
    int nb = 1; double n = 0;
    for (int z = 0; z < sl.n; z++) { B.a[z] = org_args(sl[z]); nb = std::max(nb, div_up(sl[z].n_raw, ORG_TILE)); n += sl[z].n_raw; }
    for (int z = sl.n; z < MAX_BATCH; z++) B.a[z] = B.a[0];
    const Ctx& cx = sl.first().ctx;
    LVI_LAUNCH(cx, "org_count", 20.0 * n, hipLaunchKernelGGL(org_count_kernel, dim3(nb, 1, sl.n), dim3(256), 0, cx.stream, B));
    LVI_LAUNCH(cx, "org_scan", 0, hipLaunchKernelGGL(org_scan_kernel, dim3(1, 1, sl.n), dim3(256), 0, cx.stream, B));
    LVI_LAUNCH(cx, "org_scatter", 20.0 * n + 24.0 * n, hipLaunchKernelGGL(org_scatter_kernel, dim3(nb, 1, sl.n), dim3(256), 0, cx.stream, B));
}

void stage_extract(const Slots& sl)
{
    Batch<FeatArgs> B;
    const VoxelPlan* plans[MAX_BATCH];
    double n = 0;
    for (int z = 0; z < sl.n; z++) { B.a[z] = feat_args(sl[z]); plans[z] = &sl[z].voxRing; n += sl[z].n_raw; }
    for (int z = sl.n; z < MAX_BATCH; z++) B.a[z] = B.a[0];
    LidarDev& d = sl.first();
    const Ctx& cx = d.ctx;
    LVI_LAUNCH(cx, "feat_smooth", 8.0 * n + 8.0 * n, hipLaunchKernelGGL(feat_smooth_kernel, dim3(div_up(d.ext_cap, 256), 1, sl.n), dim3(256), 0, cx.stream, B));
    LVI_LAUNCH(cx, "feat_sector", 8.0 * n, hipLaunchKernelGGL(feat_sector_kernel, dim3(d.P.N_SCAN * 6, 1, sl.n), dim3(FEAT_THREADS), 0, cx.stream, B));
    // sectors beyond the LDS-resident capacity: possible only when a ring may hold more than 6 (FEAT_SEG_CAP - 11) points
    int n_max = 0;
    for (int z = 0; z < sl.n; z++) n_max = std::max(n_max, sl[z].n_raw);
    if (std::min(d.P.Horizon_SCAN, n_max) / 6 + 12 > FEAT_SEG_CAP)
        LVI_LAUNCH(cx, "feat_sector_big", 8.0 * n, hipLaunchKernelGGL(feat_sector_big_kernel, dim3(d.P.N_SCAN, 1, sl.n), dim3(FEAT_THREADS), 0, cx.stream, B));
    LVI_LAUNCH(cx, "feat_finalize", 0, hipLaunchKernelGGL(feat_finalize_kernel, dim3(1, 1, sl.n), dim3(256), 0, cx.stream, B));
    voxel_downsample_batch(cx, plans, sl.n, "ring", n);
}

void stage_downsample(const Slots& sl)
{
    const VoxelPlan* plans[MAX_BATCH];
    double n = 0;
    for (int z = 0; z < sl.n; z++) { plans[z] = &sl[z].voxScan; n += 0.4 * sl[z].n_raw; }
    voxel_downsample_batch(sl.first().ctx, plans, sl.n, "scan", n);
}

void stage_organize(LidarDev& d) { stage_organize(OneSlot(d).s); }
void stage_extract(LidarDev& d) { stage_extract(OneSlot(d).s); }
void stage_downsample(LidarDev& d) { stage_downsample(OneSlot(d).s); }

}  // namespace lvi
