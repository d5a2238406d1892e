// pcl::VoxelGrid<PointXYZI>::applyFilter for gfx950.
//
// Per batch, all segments at once (blockIdx.y = segment), nothing returns to the host:
//   vox_init → vox_minmax (bbox, order-encoded atomics) → vox_setup (grid dims, overflow rule,
//   key width) → vox_keys (i32 voxel idx per point, exact PCL arithmetic, no FMA) →
//   stable radix sort of (idx, point index) → vox_heads_* (ordered compaction of the first
//   entry of every distinct idx) → vox_centroid (f32 running sums in sorted order, / count).
// HBM-bound: algorithmic bytes 16·P read + 16·V written; the sort adds 20·P per 8-bit pass.
#include "lvi_voxel.hpp"

namespace lvi {

namespace {

struct VoxArgs {
    const VoxSegStatic* st; const VoxSegDyn* dyn; VoxGrid* grid;
    int *d_n, *d_nbits;
    unsigned *keysA, *valsA, *keysB, *valsB;
    int* blockHeads; int* starts; int* nout;
    int nseg, seg_cap, nblk_h, concat;
    float* mmPartial; int nblk_mm;      // [nseg][nblk_mm][8]: min xyz, max xyz, count (as float bits), pad
};

__global__ void vox_init_kernel(VoxArgs a)
{
    const int s = threadIdx.x;
    if (s >= a.nseg) return;
    VoxGrid& g = a.grid[s];
    g.bb[0] = g.bb[1] = g.bb[2] = 0xFFFFFFFFu;
    g.bb[3] = g.bb[4] = g.bb[5] = 0u;
    g.n_valid = 0;
    int n = a.dyn[s].n;
    if (n < 0) n = 0;
    if (n > a.seg_cap) n = a.seg_cap;
    a.d_n[s] = n;
}

__global__ __launch_bounds__(256) void vox_minmax_kernel(VoxArgs a)
{
    const int s = blockIdx.y;
    const int n = a.d_n[s];
    const lvi_pt* __restrict__ in = a.st[s].in + a.dyn[s].in_off;
    const uint8_t* __restrict__ mask = a.st[s].mask ? a.st[s].mask + a.dyn[s].in_off : nullptr;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    const int stride = gridDim.x * 256;
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        lvi_pt p[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {               // four independent loads in flight per lane
            const int i = i0 + u * stride;
            ok[u] = i < n && (!mask || mask[i]);
            if (ok[u]) p[u] = in[i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!ok[u]) continue;
            mn[0] = fminf(mn[0], p[u].x); mn[1] = fminf(mn[1], p[u].y); mn[2] = fminf(mn[2], p[u].z);
            mx[0] = fmaxf(mx[0], p[u].x); mx[1] = fmaxf(mx[1], p[u].y); mx[2] = fmaxf(mx[2], p[u].z);
            cnt++;
        }
    }
    // workgroup reduction to one partial record per workgroup; vox_setup folds the records.  (Atomics on the
    // seven bbox words serialise: ~15 ns each, 0.1 ms for a 5M-point map with 1024 workgroups.)
    __shared__ float smn[4][3], smx[4][3];
    __shared__ int scnt[4];
    cnt = wave_sum(cnt);
#pragma unroll
    for (int d = 0; d < 3; d++) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    if (lane_id() == 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) { smn[wave_id()][d] = mn[d]; smx[wave_id()][d] = mx[d]; }
        scnt[wave_id()] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int w = 0; w < 4; w++) {
            c += scnt[w];
#pragma unroll
            for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], smn[w][d]); hi[d] = fmaxf(hi[d], smx[w][d]); }
        }
        float* rec = a.mmPartial + ((size_t)s * a.nblk_mm + blockIdx.x) * 8;
        rec[0] = lo[0]; rec[1] = lo[1]; rec[2] = lo[2]; rec[3] = hi[0]; rec[4] = hi[1]; rec[5] = hi[2];
        rec[6] = __int_as_float(c); rec[7] = 0.f;
    }
}

// grid geometry of one segment from its bbox (g.bb, g.n_valid already set): PCL's overflow rule, min_b / div_b /
// divb_mul, key width.  Shared by the multi-workgroup and the single-workgroup paths.
__device__ void vox_setup_math(VoxGrid& g, float leaf, int seg_cap)
{
    g.overflow = 0; g.nvox = 0; g.out_off = 0;
    if (g.n_valid == 0) { g.sentinel = 0u; g.nbits = 0; g.inv = 0.f; return; }
    const float inv = div_rn(1.0f, leaf);
    g.inv = inv;
    float mnp[3], mxp[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { mnp[d] = ord2f(g.bb[d]); mxp[d] = ord2f(g.bb[3 + d]); }
    // dx = static_cast<int64>((max-min)*inv) + 1 … ; dx*dy*dz > INT32_MAX.  Evaluated in double (exact for
    // every product that can pass the test; hipcc 7.2 crashes in isel on the f32→i64 form of this kernel).
    const double dx = trunc((double)mul_rn(sub_rn(mxp[0], mnp[0]), inv)) + 1.0;
    const double dy = trunc((double)mul_rn(sub_rn(mxp[1], mnp[1]), inv)) + 1.0;
    const double dz = trunc((double)mul_rn(sub_rn(mxp[2], mnp[2]), inv)) + 1.0;
    if (!(dx * dy * dz <= 2147483647.0)) {
        // "Leaf size is too small for the input dataset. Integer indices would overflow." → output = input.
        // Realised as one voxel per point: key = point index (already ascending, sort is a no-op permutation).
        g.overflow = 1;
        g.sentinel = (unsigned)seg_cap;
        g.nbits = 32 - __clz((unsigned)seg_cap);
        return;
    }
    int maxb[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        g.min_b[d] = (int)floorf(mul_rn(mnp[d], inv));
        maxb[d] = (int)floorf(mul_rn(mxp[d], inv));
        g.div_b[d] = maxb[d] - g.min_b[d] + 1;
    }
    g.mul1 = (unsigned)g.div_b[0];
    g.mul2 = (unsigned)g.div_b[0] * (unsigned)g.div_b[1];
    const unsigned long long ncells = (unsigned long long)g.div_b[0] * (unsigned long long)g.div_b[1] * (unsigned long long)g.div_b[2];
    if (ncells >= 0xFFFFFFFFull) { g.sentinel = 0xFFFFFFFFu; g.nbits = 32; }
    else {
        g.sentinel = (unsigned)ncells;
        int nb = 0;
        for (unsigned long long t = ncells; t; t >>= 1) nb++;
        g.nbits = nb;
    }
}


__global__ __launch_bounds__(64) void vox_setup_kernel(VoxArgs a)
{
    const int s = blockIdx.x;                     // one wavefront per segment
    VoxGrid& g = a.grid[s];
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        int c = 0;
        for (int b = threadIdx.x; b < a.nblk_mm; b += 64) {
            const float* rec = a.mmPartial + ((size_t)s * a.nblk_mm + b) * 8;
            const int cb = __float_as_int(rec[6]);
            if (cb > 0) {
                c += cb;
#pragma unroll
                for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], rec[d]); hi[d] = fmaxf(hi[d], rec[3 + d]); }
            }
        }
        c = wave_sum(c);
#pragma unroll
        for (int d = 0; d < 3; d++) { lo[d] = wave_min(lo[d]); hi[d] = wave_max(hi[d]); }
        if (threadIdx.x != 0) return;
        g.n_valid = c;
        if (c > 0) {
#pragma unroll
            for (int d = 0; d < 3; d++) { g.bb[d] = f2ord(lo[d]); g.bb[3 + d] = f2ord(hi[d]); }
        }
    }
    vox_setup_math(g, a.st[s].leaf, a.seg_cap);
    a.d_nbits[s] = g.nbits;
}

// PCL voxel idx of input point i of a segment: ijk = int(floor(p * inv) - float(min_b)), idx = ijk . divb_mul (i32 wrap)
__device__ __forceinline__ unsigned vox_key_of(const VoxGrid& g, const lvi_pt* in, const uint8_t* mask, int off, int i)
{
    if (mask && !mask[off + i]) return g.sentinel;
    if (g.overflow) return (unsigned)i;
    const lvi_pt p = in[off + i];
    const int ijk0 = (int)sub_rn(floorf(mul_rn(p.x, g.inv)), (float)g.min_b[0]);
    const int ijk1 = (int)sub_rn(floorf(mul_rn(p.y, g.inv)), (float)g.min_b[1]);
    const int ijk2 = (int)sub_rn(floorf(mul_rn(p.z, g.inv)), (float)g.min_b[2]);
    return (unsigned)ijk0 + (unsigned)ijk1 * g.mul1 + (unsigned)ijk2 * g.mul2;
}

__global__ __launch_bounds__(256) void vox_keys_kernel(VoxArgs a)
{
    const int s = blockIdx.y;
    const int n = a.d_n[s];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const VoxGrid& g = a.grid[s];
    const unsigned key = vox_key_of(g, a.st[s].in, a.st[s].mask, a.dyn[s].in_off, i);
    const size_t o = (size_t)s * a.seg_cap + i;
    a.keysA[o] = key;
    a.valsA[o] = (unsigned)i;
}

__device__ __forceinline__ const unsigned* sorted_keys(const VoxArgs& a, int s)
{
    return (rs_result_in_B(a.d_nbits[s]) ? a.keysB : a.keysA) + (size_t)s * a.seg_cap;
}
__device__ __forceinline__ const unsigned* sorted_vals(const VoxArgs& a, int s)
{
    return (rs_result_in_B(a.d_nbits[s]) ? a.valsB : a.valsA) + (size_t)s * a.seg_cap;
}

__device__ __forceinline__ bool is_head(const unsigned* keys, int i, unsigned sentinel)
{
    const unsigned k = keys[i];
    return k != sentinel && (i == 0 || keys[i - 1] != k);
}

__global__ __launch_bounds__(256) void vox_heads_count_kernel(VoxArgs a)
{
    const int s = blockIdx.y;
    const int n = a.d_n[s];
    const int base = blockIdx.x * VOX_HT;
    if (base >= n) return;
    const unsigned* keys = sorted_keys(a, s);
    const unsigned sent = a.grid[s].sentinel;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int i = base + threadIdx.x * 4 + j; if (i < n && is_head(keys, i, sent)) c++; }
    __shared__ int ws[8];
    int tot;
    block_excl_scan<256>(c, ws, &tot);
    if (threadIdx.x == 0) a.blockHeads[s * a.nblk_h + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void vox_heads_scan_kernel(VoxArgs a)
{
    __shared__ int ws[8];
    for (int s = 0; s < a.nseg; s++) {
        const int n = a.d_n[s];
        const int nt = (n + VOX_HT - 1) / VOX_HT;
        int* row = a.blockHeads + s * a.nblk_h;
        int carry = 0;
        for (int c = 0; c < nt; c += 256) {
            const int i = c + threadIdx.x;
            const int v = (i < nt) ? row[i] : 0;
            int tot;
            const int ex = block_excl_scan<256>(v, ws, &tot);
            if (i < nt) row[i] = carry + ex;
            carry += tot;
        }
        if (threadIdx.x == 0) a.grid[s].nvox = carry;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int off = 0;
        for (int s = 0; s < a.nseg; s++) {
            a.grid[s].out_off = a.concat ? off : 0;
            a.nout[s] = a.grid[s].nvox;
            off += a.grid[s].nvox;
        }
        a.nout[a.nseg] = off;
    }
}

__global__ __launch_bounds__(256) void vox_heads_assign_kernel(VoxArgs a)
{
    const int s = blockIdx.y;
    const int n = a.d_n[s];
    const int base = blockIdx.x * VOX_HT;
    if (base >= n) return;
    const unsigned* keys = sorted_keys(a, s);
    const unsigned sent = a.grid[s].sentinel;
    int* starts = a.starts + (size_t)s * ((size_t)a.seg_cap + 1);
    const int nvox = a.grid[s].nvox;
    bool h[4]; int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int i = base + threadIdx.x * 4 + j; h[j] = (i < n) && is_head(keys, i, sent); c += h[j]; }
    __shared__ int ws[8];
    int v = a.blockHeads[s * a.nblk_h + blockIdx.x] + block_excl_scan<256>(c, ws, nullptr);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = base + threadIdx.x * 4 + j;
        if (i >= n) break;
        if (h[j]) starts[v++] = i;
        // end of the last voxel = first sentinel entry, or n
        const unsigned k = keys[i];
        if (k == sent) { if (i == 0 || keys[i - 1] != sent) starts[nvox] = i; }
        else if (i == n - 1) starts[nvox] = n;
    }
}

// pcl::CentroidPoint: f32 sums of x,y,z,intensity over the voxel's points, divided by the count.
// VOX_CG (8, or 32 for the dense local map) lanes share a voxel: lane i sums points i, i+G, … of the voxel (sorted order = input order, the
// sort is stable), then the G partial sums are combined in lane order.  For voxels of <= G points this
// is exactly the sequential sum; for larger ones it is one more of the orders PCL's unstable sort allows.
template <int VOX_CG>
__global__ __launch_bounds__(256) void vox_centroid_kernel(VoxArgs a)
{
    const int s = blockIdx.y;
    const VoxGrid& g = a.grid[s];
    const int sub = threadIdx.x % VOX_CG;
    const int per_block = 256 / VOX_CG;
    const int nblocks_needed = (g.nvox + per_block - 1) / per_block;
    for (int blk = blockIdx.x; blk < nblocks_needed; blk += gridDim.x) {
        const int v = blk * per_block + threadIdx.x / VOX_CG;
        const bool act = v < g.nvox;
        const int* starts = a.starts + (size_t)s * ((size_t)a.seg_cap + 1);
        const unsigned* vals = sorted_vals(a, s);
        const lvi_pt* __restrict__ in = a.st[s].in + a.dyn[s].in_off;
        int b = 0, e = 0;
        if (act) { b = starts[v]; e = starts[v + 1]; }
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (int j = b + sub; j < e; j += VOX_CG) {
            const lvi_pt p = in[vals[j]];
            sx = add_rn(sx, p.x); sy = add_rn(sy, p.y); sz = add_rn(sz, p.z); si = add_rn(si, p.intensity);
        }
        // combine partials in lane order 0,1,…,G-1 (lane 0 ends with the total)
        float tx = sx, ty = sy, tz = sz, ti = si;
#pragma unroll
        for (int q = 1; q < VOX_CG; q++) {
            const float ox = __shfl_down(sx, q, VOX_CG), oy = __shfl_down(sy, q, VOX_CG), oz = __shfl_down(sz, q, VOX_CG), oi = __shfl_down(si, q, VOX_CG);
            if (b + q < e) { tx = add_rn(tx, ox); ty = add_rn(ty, oy); tz = add_rn(tz, oz); ti = add_rn(ti, oi); }
        }
        if (act && sub == 0) {
            const float cnt = (float)(e - b);
            lvi_pt o;
            o.x = div_rn(tx, cnt); o.y = div_rn(ty, cnt); o.z = div_rn(tz, cnt); o.intensity = div_rn(ti, cnt);
            lvi_pt* out = a.concat ? a.st[0].out : a.st[s].out;
            out[g.out_off + v] = o;
        }
    }
}

// =====================================================================================================
// Single-workgroup path for tiny plans (capacity <= VOX_SMALL_MAX points per segment).
// The multi-workgroup pipeline above is 20 launches; with a few thousand points each of them is
// pure launch latency.  Here ONE 1024-thread workgroup per segment does bbox → geometry → keys → radix
// sort (only as many 8-bit passes as the key width needs, ping-pong in global scratch that stays in L2)
// → ordered compaction of the voxel starts; a second small launch writes the centroids (it needs the
// voxel counts of the preceding segments for the concatenated output).  Same arithmetic, same stable
// order, therefore the same bits as the large path.
// =====================================================================================================
constexpr int VS_THREADS = 1024;
constexpr int VS_NW = VS_THREADS / 64;
constexpr int VS_ITEMS = 4;
constexpr int VS_TILE = VS_THREADS * VS_ITEMS;

__global__ __launch_bounds__(VS_THREADS) void vox_small_kernel(VoxArgs a)
{
    const int s = blockIdx.x, tid = threadIdx.x, w = wave_id(), l = lane_id();
    __shared__ VoxGrid sg;
    __shared__ float smn[VS_NW][3], smx[VS_NW][3];
    __shared__ int scnt[VS_NW];
    __shared__ unsigned hist[256], dbase[256];
    __shared__ unsigned waveCnt[VS_NW][256];
    __shared__ int ws[VS_NW + 2];
    int n = a.dyn[s].n;
    n = n < 0 ? 0 : (n > a.seg_cap ? a.seg_cap : n);
    const int off = a.dyn[s].in_off;
    const lvi_pt* __restrict__ in = a.st[s].in;
    const uint8_t* __restrict__ mask = a.st[s].mask;
    // ---- bbox
    {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        int cnt = 0;
        for (int i = tid; i < n; i += VS_THREADS) {
            if (mask && !mask[off + i]) continue;
            const lvi_pt p = in[off + i];
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
            cnt++;
        }
        cnt = wave_sum(cnt);
#pragma unroll
        for (int d = 0; d < 3; d++) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
        if (l == 0) {
#pragma unroll
            for (int d = 0; d < 3; d++) { smn[w][d] = mn[d]; smx[w][d] = mx[d]; }
            scnt[w] = cnt;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int q = 0; q < VS_NW; q++) {
            c += scnt[q];
#pragma unroll
            for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], smn[q][d]); hi[d] = fmaxf(hi[d], smx[q][d]); }
        }
        sg.n_valid = c;
#pragma unroll
        for (int d = 0; d < 3; d++) { sg.bb[d] = c > 0 ? f2ord(lo[d]) : 0xFFFFFFFFu; sg.bb[3 + d] = c > 0 ? f2ord(hi[d]) : 0u; }
        vox_setup_math(sg, a.st[s].leaf, a.seg_cap);
        a.d_n[s] = n; a.d_nbits[s] = sg.nbits;
    }
    __syncthreads();
    const size_t so = (size_t)s * a.seg_cap;
    // ---- keys
    for (int i = tid; i < n; i += VS_THREADS) { a.keysA[so + i] = vox_key_of(sg, in, mask, off, i); a.valsA[so + i] = (unsigned)i; }
    __syncthreads();
    // ---- LSD radix sort, ceil(nbits/8) passes
    const int npass = (sg.nbits + 7) >> 3;
    const uint64_t lt = lanemask_lt();
    for (int pass = 0; pass < npass; pass++) {
        const int shift = pass * 8;
        const unsigned* __restrict__ srcK = ((pass & 1) ? a.keysB : a.keysA) + so;
        const unsigned* __restrict__ srcV = ((pass & 1) ? a.valsB : a.valsA) + so;
        unsigned* __restrict__ dstK = ((pass & 1) ? a.keysA : a.keysB) + so;
        unsigned* __restrict__ dstV = ((pass & 1) ? a.valsA : a.valsB) + so;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += VS_THREADS) atomicAdd(&hist[(srcK[i] >> shift) & 255u], 1u);
        __syncthreads();
        {
            int tot;
            const int ex = block_excl_scan<VS_THREADS>(tid < 256 ? (int)hist[tid] : 0, ws, &tot);
            if (tid < 256) dbase[tid] = (unsigned)ex;                  // running global base of every digit
        }
        __syncthreads();
        for (int base = 0; base < n; base += VS_TILE) {
#pragma unroll
            for (int q = 0; q < VS_NW; q += 4) if (tid < 256) { waveCnt[q][tid] = 0; waveCnt[q + 1][tid] = 0; waveCnt[q + 2][tid] = 0; waveCnt[q + 3][tid] = 0; }
            __syncthreads();
            unsigned k[VS_ITEMS], v[VS_ITEMS]; unsigned short r[VS_ITEMS];
            const int cbase = base + w * (VS_ITEMS * 64);
#pragma unroll
            for (int i = 0; i < VS_ITEMS; i++) {
                const int idx = cbase + i * 64 + l;
                const bool valid = idx < n;
                k[i] = valid ? srcK[idx] : 0u; v[i] = valid ? srcV[idx] : 0u;
                const unsigned d = (k[i] >> shift) & 255u;
                uint64_t peers = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 8; b++) { const bool bit = (d >> b) & 1u; const uint64_t m = __ballot(bit); peers &= bit ? m : ~m; }
                r[i] = 0;
                if (valid) {
                    const unsigned prior = waveCnt[w][d];
                    r[i] = (unsigned short)(prior + __popcll(peers & lt));
                    if ((peers & lt) == 0) waveCnt[w][d] = prior + __popcll(peers);
                }
            }
            __syncthreads();
            unsigned tileCnt = 0;
            if (tid < 256) {                                           // per digit: offsets of the waves inside the tile
#pragma unroll
                for (int q = 0; q < VS_NW; q++) { const unsigned c = waveCnt[q][tid]; waveCnt[q][tid] = tileCnt; tileCnt += c; }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VS_ITEMS; i++) {
                const int idx = cbase + i * 64 + l;
                if (idx < n) {
                    const unsigned d = (k[i] >> shift) & 255u;
                    const unsigned g = dbase[d] + waveCnt[w][d] + r[i];
                    dstK[g] = k[i]; dstV[g] = v[i];
                }
            }
            __syncthreads();
            if (tid < 256) dbase[tid] += tileCnt;
            __syncthreads();
        }
    }
    // ---- voxel starts: ordered compaction of the first entry of every distinct key
    const unsigned* keys = ((npass & 1) ? a.keysB : a.keysA) + so;
    int* starts = a.starts + (size_t)s * ((size_t)a.seg_cap + 1);
    const unsigned sent = sg.sentinel;
    int carry = 0, first_sent = n;
    for (int base = 0; base < n; base += VS_TILE) {
        bool h[VS_ITEMS]; int c = 0;
#pragma unroll
        for (int j = 0; j < VS_ITEMS; j++) { const int i = base + tid * VS_ITEMS + j; h[j] = (i < n) && is_head(keys, i, sent); c += h[j]; }
        int tot;
        int vv = carry + block_excl_scan<VS_THREADS>(c, ws, &tot);
#pragma unroll
        for (int j = 0; j < VS_ITEMS; j++) {
            const int i = base + tid * VS_ITEMS + j;
            if (i >= n) break;
            if (h[j]) starts[vv++] = i;
            if (keys[i] == sent && (i == 0 || keys[i - 1] != sent)) first_sent = i;
        }
        carry += tot;
    }
    // end of the last voxel = first masked-out entry (at most one thread found it), else n
    __shared__ int s_end;
    if (tid == 0) s_end = n;
    __syncthreads();
    if (first_sent < n) s_end = first_sent;
    __syncthreads();
    if (tid == 0) {
        starts[carry] = s_end;
        sg.nvox = carry;
        a.grid[s] = sg;
    }
}

// out offsets of the concatenated output + centroids, 8 lanes per voxel (as vox_centroid_kernel<8>)
__global__ __launch_bounds__(VS_THREADS) void vox_small_finish_kernel(VoxArgs a)
{
    const int s = blockIdx.x;
    __shared__ int s_off;
    if (threadIdx.x == 0) {
        int off = 0, total = 0;
        for (int q = 0; q < a.nseg; q++) { if (q < s) off += a.grid[q].nvox; total += a.grid[q].nvox; }
        s_off = a.concat ? off : 0;
        a.grid[s].out_off = s_off;
        a.nout[s] = a.grid[s].nvox;
        if (s == 0) a.nout[a.nseg] = total;
    }
    __syncthreads();
    const int out_off = s_off;
    const int nvox = a.grid[s].nvox;
    constexpr int G = 8;
    const int sub = threadIdx.x % G;
    const int* starts = a.starts + (size_t)s * ((size_t)a.seg_cap + 1);
    const unsigned* vals = sorted_vals(a, s);
    const lvi_pt* __restrict__ in = a.st[s].in + a.dyn[s].in_off;
    lvi_pt* out = a.concat ? a.st[0].out : a.st[s].out;
    for (int v0 = 0; v0 < nvox; v0 += VS_THREADS / G) {
        const int v = v0 + threadIdx.x / G;
        const bool act = v < nvox;
        int b = 0, e = 0;
        if (act) { b = starts[v]; e = starts[v + 1]; }
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (int j = b + sub; j < e; j += G) {
            const lvi_pt p = in[vals[j]];
            sx = add_rn(sx, p.x); sy = add_rn(sy, p.y); sz = add_rn(sz, p.z); si = add_rn(si, p.intensity);
        }
        float tx = sx, ty = sy, tz = sz, ti = si;
#pragma unroll
        for (int q = 1; q < G; q++) {
            const float ox = __shfl_down(sx, q, G), oy = __shfl_down(sy, q, G), oz = __shfl_down(sz, q, G), oi = __shfl_down(si, q, G);
            if (b + q < e) { tx = add_rn(tx, ox); ty = add_rn(ty, oy); tz = add_rn(tz, oz); ti = add_rn(ti, oi); }
        }
        if (act && sub == 0) {
            const float cnt = (float)(e - b);
            lvi_pt o;
            o.x = div_rn(tx, cnt); o.y = div_rn(ty, cnt); o.z = div_rn(tz, cnt); o.intensity = div_rn(ti, cnt);
            out[out_off + v] = o;
        }
    }
}

}  // namespace

void VoxelPlan::set_static(const Ctx& ctx, const VoxSegStatic* host_segs) const
{
    LVI_HIP(hipMemcpyAsync(d_static, host_segs, sizeof(VoxSegStatic) * nseg, hipMemcpyHostToDevice, ctx.stream));
    LVI_HIP(hipStreamSynchronize(ctx.stream));
}

// Debug view of segment 0 of the last run (tests only): per-input-point keys, distinct keys in
// output order and points per output voxel.  Overflow-rule runs report empty arrays.
void voxel_debug_fetch(const Ctx& ctx, const VoxelPlan& p, int n_in, std::vector<int32_t>& keys, std::vector<int32_t>& cells, std::vector<int32_t>& counts)
{
    keys.clear(); cells.clear(); counts.clear();
    VoxGrid g;
    LVI_HIP(hipMemcpyAsync(&g, p.d_grid, sizeof(g), hipMemcpyDeviceToHost, ctx.stream));
    LVI_HIP(hipStreamSynchronize(ctx.stream));
    if (g.overflow || n_in <= 0 || g.n_valid == 0) return;
    const bool inB = (((g.nbits + 7) >> 3) & 1) != 0;
    std::vector<unsigned> sk(n_in);
    std::vector<int> st(g.nvox + 1);
    LVI_HIP(hipMemcpyAsync(sk.data(), inB ? p.sort.keysB : p.sort.keysA, sizeof(unsigned) * n_in, hipMemcpyDeviceToHost, ctx.stream));
    LVI_HIP(hipMemcpyAsync(st.data(), p.d_starts, sizeof(int) * (g.nvox + 1), hipMemcpyDeviceToHost, ctx.stream));
    LVI_HIP(hipStreamSynchronize(ctx.stream));
    cells.resize(g.nvox); counts.resize(g.nvox);
    for (int v = 0; v < g.nvox; v++) { cells[v] = (int32_t)sk[st[v]]; counts[v] = st[v + 1] - st[v]; }
    // per-point keys: recompute into keysA (scratch is free once the outputs are written)
    VoxArgs a{p.d_static, p.d_dyn, p.d_grid, p.d_n, p.d_nbits, p.sort.keysA, p.sort.valsA, p.sort.keysB, p.sort.valsB,
              p.d_blockHeads, p.d_starts, p.d_nout, p.nseg, p.seg_cap, p.nblk_h, p.concat_out ? 1 : 0, p.d_mmPartial, p.nblk_mm};
    hipLaunchKernelGGL(vox_keys_kernel, dim3(div_up(n_in, 256), 1), dim3(256), 0, ctx.stream, a);
    LVI_HIP(hipGetLastError());
    keys.resize(n_in);
    LVI_HIP(hipMemcpyAsync(keys.data(), p.sort.keysA, sizeof(unsigned) * n_in, hipMemcpyDeviceToHost, ctx.stream));
    LVI_HIP(hipStreamSynchronize(ctx.stream));
}

void voxel_downsample_batch(const Ctx& ctx, const VoxelPlan& p, const char* tag, double n_hint)
{
    VoxArgs a{p.d_static, p.d_dyn, p.d_grid, p.d_n, p.d_nbits, p.sort.keysA, p.sort.valsA, p.sort.keysB, p.sort.valsB,
              p.d_blockHeads, p.d_starts, p.d_nout, p.nseg, p.seg_cap, p.nblk_h, p.concat_out ? 1 : 0, p.d_mmPartial, p.nblk_mm};
    if (p.seg_cap <= VOX_SMALL_MAX) {
        char n0[48], n1[48];
        snprintf(n0, sizeof(n0), "vox_small/%s", tag); snprintf(n1, sizeof(n1), "vox_small_finish/%s", tag);
        LVI_LAUNCH(ctx, n0, 16.0 * n_hint, hipLaunchKernelGGL(vox_small_kernel, dim3(p.nseg), dim3(VS_THREADS), 0, ctx.stream, a));
        LVI_LAUNCH(ctx, n1, 20.0 * n_hint, hipLaunchKernelGGL(vox_small_finish_kernel, dim3(p.nseg), dim3(VS_THREADS), 0, ctx.stream, a));
        return;
    }
    char nm[8][48];
    const char* base[8] = {"vox_init", "vox_minmax", "vox_setup", "vox_keys", "vox_heads_count", "vox_heads_scan", "vox_heads_assign", "vox_centroid"};
    for (int i = 0; i < 8; i++) snprintf(nm[i], sizeof(nm[i]), "%s/%s", base[i], tag);
    const int mm_blocks = p.nblk_mm;
    const dim3 gp(div_up(p.seg_cap, 256), p.nseg), gh(p.nblk_h, p.nseg);
    LVI_LAUNCH(ctx, nm[0], 0, hipLaunchKernelGGL(vox_init_kernel, dim3(1), dim3(64), 0, ctx.stream, a));
    LVI_LAUNCH(ctx, nm[1], 16.0 * n_hint, hipLaunchKernelGGL(vox_minmax_kernel, dim3(mm_blocks, p.nseg), dim3(256), 0, ctx.stream, a));
    LVI_LAUNCH(ctx, nm[2], 0, hipLaunchKernelGGL(vox_setup_kernel, dim3(p.nseg), dim3(64), 0, ctx.stream, a));
    LVI_LAUNCH(ctx, nm[3], 24.0 * n_hint, hipLaunchKernelGGL(vox_keys_kernel, gp, dim3(256), 0, ctx.stream, a));
    radix_sort_pairs(ctx, p.sort, p.d_n, p.d_nbits, 4, tag, n_hint);
    LVI_LAUNCH(ctx, nm[4], 4.0 * n_hint, hipLaunchKernelGGL(vox_heads_count_kernel, gh, dim3(256), 0, ctx.stream, a));
    LVI_LAUNCH(ctx, nm[5], 0, hipLaunchKernelGGL(vox_heads_scan_kernel, dim3(1), dim3(256), 0, ctx.stream, a));
    LVI_LAUNCH(ctx, nm[6], 4.0 * n_hint, hipLaunchKernelGGL(vox_heads_assign_kernel, gh, dim3(256), 0, ctx.stream, a));
    const dim3 gc(std::max(1, std::min(div_up(p.seg_cap, 256 / 8), 8192)), p.nseg);
    if (p.centroid_lanes >= 32)
        LVI_LAUNCH(ctx, nm[7], 20.0 * n_hint, hipLaunchKernelGGL(vox_centroid_kernel<32>, gc, dim3(256), 0, ctx.stream, a));
    else
        LVI_LAUNCH(ctx, nm[7], 20.0 * n_hint, hipLaunchKernelGGL(vox_centroid_kernel<8>, gc, dim3(256), 0, ctx.stream, a));
}

}  // namespace lvi
