// pcl::VoxelGrid<PointXYZI>::applyFilter for gfx950.
//
// Per batch, all segments at once (blockIdx.y = segment), nothing returns to the host:
//   vox_minmax (bbox partials per workgroup) → vox_setup (grid dims, overflow rule, key width, bins,
//   fixed-point scales), then one of
//   SORTED: vox_keys (i32 voxel idx per point, exact PCL arithmetic, no FMA) → stable radix sort of
//           (idx, point index) → vox_heads_* (ordered compaction of the first entry of every distinct
//           idx) → vox_centroid (gathers the voxel's points);
//   BINNED: vb_hist → vb_scan → vb_scatter (partition of the POINTS into bins of consecutive idx,
//           LDS histogram + one global reservation per occupied bin and tile) → vb_accum (one
//           workgroup per bin, LDS-resident accumulators, ordered compaction of the occupied
//           voxels) → vb_outscan → vb_copy.
// Voxel idx, voxel set and output order are PCL's, bit for bit.  The centroid is the exact mean of the
// voxel's points in fixed point (integer sums, so independent of visiting order and identical in both
// paths), rounded once to f32; PCL sums in f32 in the order its unstable sort leaves the points, which
// no parallel machine reproduces — the difference is below 1e-5 m and covered by the tests' tolerance.
// HBM-bound: algorithmic bytes 16·P read + 16·V written.
#include "lvi_voxel.hpp"

namespace lvi {

namespace {

struct VoxArgs {
    const VoxSegStatic* st; const VoxSegDyn* dyn; VoxGrid* grid;
    int *d_n, *d_nbits;
    unsigned *keysA, *valsA, *keysB, *valsB;
    int* blockHeads; int* starts; int* nout;
    int nseg, seg_cap, nblk_h, concat;
    float* mmPartial; int nblk_mm;      // [nseg][nblk_mm][12]: min xyz, max xyz, count (as float bits), min/max intensity
    unsigned* binCount; int* binStart; unsigned* cursor; int* binVox; int* binOut;
    lvi_pt* bucketed; lvi_pt* staging; uint2* stagingKC;
    unsigned long long* h_ncells;
    int* chunkStart; int* multiStart;                  // [nseg][VB_NB+1] exclusive scans of chunks per bin / chunks of multi-chunk bins
    int* chunkBin; int max_chunks;                     // [nseg][max_chunks]
    int* lightBin;                                     // [nseg][VB_NB + 1]
    int* multiOwner;                                   // [nseg][max_multi] chunk-table slot -> its bin if the slot is the bin's first, else -1
    unsigned long long* chunkTabV; unsigned* chunkTabC; int max_multi;   // [nseg][max_multi] LDS tables of the chunks of multi-chunk bins
    int n_host[4]; int use_n_host;                     // host-known segment lengths (raw map), else dyn[].n
    const int* n_dev[4];                               // producer's device counters (scan grids), else dyn[].n
    int bin_pts, bin_max;                              // binned path: aim at bin_pts points per bin, at most bin_max bins
    const unsigned* binCountCached;                    // [nseg][VB_NB] per-bin point counts from vb_colscan (plans with a deterministic partition): vb_hist is skipped
    unsigned* binCountOut;                             // … where vb_colscan writes them
    int* planMiss;                                     // [nseg] vb_plan: a point outside the previous run's grid
    int plan_spec;                                     // this run takes bbox and per-bin counts in one pass (vb_plan); vox_setup validates the counts
    unsigned* wprefix;                                 // [nseg][VB_WG][VB_NB] points of bin b in the ranges of workgroups < w (deterministic partition)
    int ch;                                            // points per accumulate chunk of this launch sequence: VB_CH, or a multiple of it when many slots fill the chip anyway (<= 32768: chunk-table counts are 16 bits)
};

__device__ __forceinline__ int seg_len(const VoxArgs& a, int s)
{
    const int n = a.use_n_host ? a.n_host[s] : ((s < 4 && a.n_dev[s]) ? *a.n_dev[s] : a.dyn[s].n);
    return n < 0 ? 0 : (n > a.seg_cap ? a.seg_cap : n);
}

__global__ __launch_bounds__(256) void vox_minmax_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y;
    const int n = seg_len(a, s);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.d_n[s] = n;          // read by every later kernel of the batch
    const lvi_pt* __restrict__ in = a.st[s].in + a.dyn[s].in_off;
    const uint8_t* __restrict__ mask = a.st[s].mask ? a.st[s].mask + a.dyn[s].in_off : nullptr;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float imn = INFINITY, imx = -INFINITY;
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
            imn = fminf(imn, p[u].intensity); imx = fmaxf(imx, p[u].intensity);
            cnt++;
        }
    }
    // workgroup reduction to one partial record per workgroup; vox_setup folds the records.  (Atomics on the
    // seven bbox words serialise: ~15 ns each, 0.1 ms for a 5M-point map with 1024 workgroups.)
    __shared__ float smn[4][4], smx[4][4];
    __shared__ int scnt[4];
    cnt = wave_sum(cnt);
#pragma unroll
    for (int d = 0; d < 3; d++) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    imn = wave_min(imn); imx = wave_max(imx);
    if (lane_id() == 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) { smn[wave_id()][d] = mn[d]; smx[wave_id()][d] = mx[d]; }
        smn[wave_id()][3] = imn; smx[wave_id()][3] = imx;
        scnt[wave_id()] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int w = 0; w < 4; w++) {
            c += scnt[w];
#pragma unroll
            for (int d = 0; d < 4; d++) { lo[d] = fminf(lo[d], smn[w][d]); hi[d] = fmaxf(hi[d], smx[w][d]); }
        }
        float* rec = a.mmPartial + ((size_t)s * a.nblk_mm + blockIdx.x) * 12;
        rec[0] = lo[0]; rec[1] = lo[1]; rec[2] = lo[2]; rec[3] = hi[0]; rec[4] = hi[1]; rec[5] = hi[2];
        rec[6] = __int_as_float(c); rec[7] = lo[3]; rec[8] = hi[3];
    }
}

// fixed-point scales of the centroid sums (VoxGrid): from the leaf size and the intensity range only
__device__ void vox_fx_setup(VoxGrid& g, float leaf, float ilo, float ihi)
{
    int ex = 0, exi = 0;
    (void)frexpf(2.0f * leaf, &ex);                                   // |value - cell * leaf| < 2 leaf <= 2^ex
    (void)frexpf(fmaxf(fabsf(ilo), fabsf(ihi)), &exi);                // |intensity| < 2^exi
    g.leaf_d = (double)leaf;
    g.fx_k = 37 - ex;                                                 // 2^25 points cannot overflow 63 bits
    g.fx_ki = 37 - max(exi, 8);
}

// grid geometry of one segment from its bbox (g.bb, g.n_valid already set): PCL's overflow rule, min_b / div_b /
// divb_mul, key width.  Shared by the multi-workgroup and the single-workgroup paths.
__device__ void vox_setup_math(VoxGrid& g, float leaf, int seg_cap, float ilo, float ihi, int bin_pts, int bin_max)
{
    g.overflow = 0; g.nvox = 0; g.out_off = 0;
    g.ncells = 0ull; g.nbins = 0; g.bin_shift = VB_CL_LOG;
    vox_fx_setup(g, leaf, ilo, ihi);
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
    g.ncells = ncells;
    // about a thousand bins for a dense map (fewer for a few thousand points): few enough that a tile of consecutive points touches few of them (one global reservation
    // per tile and bin), many enough to fill the chip; never more than VB_NB, and beyond 1024 voxels per bin (sparse
    // grids) a bin is swept once per occupied 1024-voxel sub-range
    const unsigned long long target = (unsigned long long)max(64, min(g.n_valid / bin_pts, bin_max));
    int sh = 6;
    while (sh < VB_CL_LOG && (ncells >> sh) > target) sh++;
    while (((ncells + (1ull << sh) - 1ull) >> sh) > (unsigned long long)VB_NB) sh++;
    g.bin_shift = sh;
    g.nbins = (int)((ncells + (1ull << sh) - 1ull) >> sh);
    if (ncells >= 0xFFFFFFFFull) { g.sentinel = 0xFFFFFFFFu; g.nbits = 32; }
    else {
        g.sentinel = (unsigned)ncells;
        int nb = 0;
        for (unsigned long long t = ncells; t; t >>= 1) nb++;
        g.nbits = nb;
    }
}


__global__ __launch_bounds__(64) void vox_setup_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.x;                     // one wavefront per segment
    VoxGrid& g = a.grid[s];
    if (threadIdx.x == 0) a.d_n[s] = seg_len(a, s);                  // (also written by vox_minmax; a batch with a cached bbox skips that pass)
    // the geometry the previous run left (vb_plan took its histogram under it)
    const int o_min0 = g.min_b[0], o_min1 = g.min_b[1], o_min2 = g.min_b[2], o_div0 = g.div_b[0], o_div1 = g.div_b[1], o_div2 = g.div_b[2];
    const int o_shift = g.bin_shift, o_nbins = g.nbins, o_over = g.overflow;
    const float o_inv = g.inv;
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float ilo = INFINITY, ihi = -INFINITY;
        int c = 0;
        const int nb_used = a.nblk_mm;
        for (int b0 = threadIdx.x; b0 < nb_used; b0 += 64 * 4) {              // four records per lane in flight (min / max / integer sum: any order)
            float r[4][9];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float* rec = a.mmPartial + ((size_t)s * a.nblk_mm + min(b0 + 64 * u, nb_used - 1)) * 12;
#pragma unroll
                for (int q = 0; q < 9; q++) r[u][q] = rec[q];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int cb = b0 + 64 * u < nb_used ? __float_as_int(r[u][6]) : 0;
                if (cb > 0) {
                    c += cb;
#pragma unroll
                    for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], r[u][d]); hi[d] = fmaxf(hi[d], r[u][3 + d]); }
                    ilo = fminf(ilo, r[u][7]); ihi = fmaxf(ihi, r[u][8]);
                }
            }
        }
        c = wave_sum(c);
#pragma unroll
        for (int d = 0; d < 3; d++) { lo[d] = wave_min(lo[d]); hi[d] = wave_max(hi[d]); }
        ilo = wave_min(ilo); ihi = wave_max(ihi);
        if (threadIdx.x != 0) return;
        g.n_valid = c;
        if (c > 0) {
#pragma unroll
            for (int d = 0; d < 3; d++) { g.bb[d] = f2ord(lo[d]); g.bb[3 + d] = f2ord(hi[d]); }
        } else {
            g.bb[0] = g.bb[1] = g.bb[2] = 0xFFFFFFFFu; g.bb[3] = g.bb[4] = g.bb[5] = 0u;
            ilo = ihi = 0.f;
        }
        vox_setup_math(g, a.st[s].leaf, a.seg_cap, ilo, ihi, a.bin_pts, a.bin_max);
        g.plan_ok = (a.plan_spec && a.planMiss[s] == 0 && !o_over && !g.overflow && o_nbins > 0 && o_nbins == g.nbins && o_shift == g.bin_shift && o_inv == g.inv &&
                     o_min0 == g.min_b[0] && o_min1 == g.min_b[1] && o_min2 == g.min_b[2] && o_div0 == g.div_b[0] && o_div1 == g.div_b[1] && o_div2 == g.div_b[2]) ? 1 : 0;
    }
    a.d_nbits[s] = g.nbits;
    if (a.h_ncells) a.h_ncells[s] = g.ncells;                        // pinned host: AUTO's hint for the next batch
}

// PCL voxel idx of input point i of a segment: ijk = int(floor(p * inv) - float(min_b)), idx = ijk . divb_mul (i32 wrap)
__device__ __forceinline__ unsigned vox_key_of_pt(const VoxGrid& g, const lvi_pt& p)
{
    const int ijk0 = (int)sub_rn(floorf(mul_rn(p.x, g.inv)), (float)g.min_b[0]);
    const int ijk1 = (int)sub_rn(floorf(mul_rn(p.y, g.inv)), (float)g.min_b[1]);
    const int ijk2 = (int)sub_rn(floorf(mul_rn(p.z, g.inv)), (float)g.min_b[2]);
    return (unsigned)ijk0 + (unsigned)ijk1 * g.mul1 + (unsigned)ijk2 * g.mul2;
}
// … from values held in registers: the grid record lives in global memory, and behind an LDS atomic or a store the compiler reads
// inv / min_b / mul again — one L2 round trip per POINT in the ISA of the partition kernels of round 2
struct VoxKeyK { float inv, mb0, mb1, mb2; unsigned mul1, mul2; };
__device__ __forceinline__ VoxKeyK vox_keyk_of(const VoxGrid& g)
{
    VoxKeyK k;
    k.inv = g.inv; k.mb0 = (float)g.min_b[0]; k.mb1 = (float)g.min_b[1]; k.mb2 = (float)g.min_b[2]; k.mul1 = g.mul1; k.mul2 = g.mul2;
    return k;
}
__device__ __forceinline__ unsigned vox_key_k(const VoxKeyK& k, const lvi_pt& p)
{
    const int ijk0 = (int)sub_rn(floorf(mul_rn(p.x, k.inv)), k.mb0);
    const int ijk1 = (int)sub_rn(floorf(mul_rn(p.y, k.inv)), k.mb1);
    const int ijk2 = (int)sub_rn(floorf(mul_rn(p.z, k.inv)), k.mb2);
    return (unsigned)ijk0 + (unsigned)ijk1 * k.mul1 + (unsigned)ijk2 * k.mul2;
}
// eight points of a lane: every load issued before the first use (clamped index, lanes masked afterwards; the mask bytes in a
// second round under a uniform test)
template <int NL, int STRIDE>
__device__ __forceinline__ void vox_load_pts(const lvi_pt* __restrict__ in, const uint8_t* __restrict__ mask, int i_first, int i_end, lvi_pt p[NL], bool keep[NL])
{
#pragma unroll
    for (int u = 0; u < NL; u++) { const int i = i_first + u * STRIDE; keep[u] = i < i_end; p[u] = ld_global_pt(in + max(min(i, i_end - 1), 0)); }
    if (mask) {
        uint8_t mk[NL];
#pragma unroll
        for (int u = 0; u < NL; u++) mk[u] = ld_global_u8(mask + max(min(i_first + u * STRIDE, i_end - 1), 0));
#pragma unroll
        for (int u = 0; u < NL; u++) keep[u] = keep[u] && mk[u] != 0;
    }
}
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

__global__ __launch_bounds__(256) void vox_keys_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
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

__global__ __launch_bounds__(256) void vox_heads_count_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
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

__global__ __launch_bounds__(256) void vox_heads_scan_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
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

__global__ __launch_bounds__(256) void vox_heads_assign_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
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

// Fixed-point image of one coordinate relative to its voxel's own origin / the exact mean back in f32 (VoxGrid, lvi_voxel.hpp)
// round-to-nearest-even of |x| < 2^51 to an integer: one f64 add (the sum's ulp is 1) instead of the ~20 instructions of the
// general f64 -> i64 conversion, which gfx950 does not have; the same value for every such x
__device__ __forceinline__ long long d2ll_rn_small(double x)
{
    const double M = 6755399441055744.0;                            // 2^52 + 2^51
    return __double_as_longlong(x + M) - __double_as_longlong(M);
}
__device__ __forceinline__ unsigned long long fx_xyz(float v, int cell, double leaf, int k)
{
    return (unsigned long long)d2ll_rn_small(ldexp((double)v - (double)cell * leaf, k));        // |.| < 2^38 (vox_fx_setup); two's complement: sums wrap correctly
}
__device__ __forceinline__ unsigned long long fx_int(float v, int k) { return (unsigned long long)d2ll_rn_small(ldexp((double)v, k)); }
// The same two images for the accumulate loops, from values held in registers: the grid record lives in global memory and the
// compiler re-reads leaf_d / fx_k behind every LDS atomic (one L2 round trip per POINT in the ISA of round 2's loop).  The scale is a
// power of two, so r * 2^k is exact and fma(r, 2^k, M) rounds exactly once, like ldexp(r, k) + M: the same integer; cellf is the
// float floor itself (integer-valued, |.| < 2^31: the same double as the int conversion gives).
struct VoxFx { float inv, mb0, mb1, mb2; unsigned mul1, mul2; double leaf, sc, sci; };
__device__ __forceinline__ VoxFx vox_fx_of(const VoxGrid& g)
{
    VoxFx f;
    f.inv = g.inv; f.mb0 = (float)g.min_b[0]; f.mb1 = (float)g.min_b[1]; f.mb2 = (float)g.min_b[2]; f.mul1 = g.mul1; f.mul2 = g.mul2;
    f.leaf = g.leaf_d; f.sc = ldexp(1.0, g.fx_k); f.sci = ldexp(1.0, g.fx_ki);
    return f;
}
__device__ __forceinline__ unsigned long long fx_rn_scaled(double r, double sc)
{
    const double M = 6755399441055744.0;
    return (unsigned long long)(__double_as_longlong(fma(r, sc, M)) - __double_as_longlong(M));
}
// voxel idx (as vox_key_of_pt) and the four fixed-point terms of one point
__device__ __forceinline__ unsigned vox_fx_point(const VoxFx& f, const lvi_pt& p, unsigned long long v[4])
{
    const float c0 = floorf(mul_rn(p.x, f.inv)), c1 = floorf(mul_rn(p.y, f.inv)), c2 = floorf(mul_rn(p.z, f.inv));
    const int ijk0 = (int)sub_rn(c0, f.mb0), ijk1 = (int)sub_rn(c1, f.mb1), ijk2 = (int)sub_rn(c2, f.mb2);
    v[0] = fx_rn_scaled((double)p.x - (double)c0 * f.leaf, f.sc);
    v[1] = fx_rn_scaled((double)p.y - (double)c1 * f.leaf, f.sc);
    v[2] = fx_rn_scaled((double)p.z - (double)c2 * f.leaf, f.sc);
    v[3] = fx_rn_scaled((double)p.intensity, f.sci);
    return (unsigned)ijk0 + (unsigned)ijk1 * f.mul1 + (unsigned)ijk2 * f.mul2;
}
__device__ __forceinline__ float fx_mean_xyz(unsigned long long sum, unsigned cnt, int cell, double leaf, int k)
{
    return (float)((double)cell * leaf + ldexp(__ll2double_rn((long long)sum) / (double)cnt, -k));
}
__device__ __forceinline__ float fx_mean_int(unsigned long long sum, unsigned cnt, int k) { return (float)ldexp(__ll2double_rn((long long)sum) / (double)cnt, -k); }
// absolute integer coordinates of a point's voxel: floor(p * inv) — the float floor PCL takes before it subtracts min_b
__device__ __forceinline__ void vox_cell_abs(const VoxGrid& g, const lvi_pt& p, int c[3])
{
    c[0] = (int)floorf(mul_rn(p.x, g.inv)); c[1] = (int)floorf(mul_rn(p.y, g.inv)); c[2] = (int)floorf(mul_rn(p.z, g.inv));
}
// … of the voxel with linear idx `key` of this grid
__device__ __forceinline__ void vox_cell_of_key(const VoxGrid& g, unsigned key, int c[3])
{
    const unsigned d0 = (unsigned)g.div_b[0], d1 = (unsigned)g.div_b[1];
    const unsigned q = key / d0;
    c[0] = (int)(key - q * d0) + g.min_b[0]; c[1] = (int)(q % d1) + g.min_b[1]; c[2] = (int)(q / d1) + g.min_b[2];
}
__device__ __forceinline__ lvi_pt fx_centroid(const VoxGrid& g, unsigned key, unsigned long long sx, unsigned long long sy, unsigned long long sz,
                                              unsigned long long si, unsigned cnt)
{
    int c[3];
    vox_cell_of_key(g, key, c);
    lvi_pt o;
    o.x = fx_mean_xyz(sx, cnt, c[0], g.leaf_d, g.fx_k); o.y = fx_mean_xyz(sy, cnt, c[1], g.leaf_d, g.fx_k);
    o.z = fx_mean_xyz(sz, cnt, c[2], g.leaf_d, g.fx_k); o.intensity = fx_mean_int(si, cnt, g.fx_ki);
    return o;
}

// SORTED path centroid (pcl::CentroidPoint over the voxel's points).  VOX_CG (8, or 32 for the dense local map)
// lanes share a voxel: lane i sums points i, i+G, … of the voxel, then the partial sums are added up.
template <int VOX_CG>
__global__ __launch_bounds__(256) void vox_centroid_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
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
        if (g.overflow) {                                   // PCL: output = input, untouched
            if (act && sub == 0) { lvi_pt* out = a.concat ? a.st[0].out : a.st[s].out; out[g.out_off + v] = in[vals[b]]; }
            continue;
        }
        unsigned long long sx = 0, sy = 0, sz = 0, si = 0;
        const unsigned vkey = act ? sorted_keys(a, s)[b] : 0u;
        int cell[3];
        vox_cell_of_key(g, vkey, cell);
        for (int j = b + sub; j < e; j += VOX_CG) {
            const lvi_pt p = in[vals[j]];
            sx += fx_xyz(p.x, cell[0], g.leaf_d, g.fx_k); sy += fx_xyz(p.y, cell[1], g.leaf_d, g.fx_k);
            sz += fx_xyz(p.z, cell[2], g.leaf_d, g.fx_k); si += fx_int(p.intensity, g.fx_ki);
        }
#pragma unroll
        for (int q = VOX_CG / 2; q > 0; q >>= 1) {
            sx += __shfl_down(sx, q, VOX_CG); sy += __shfl_down(sy, q, VOX_CG); sz += __shfl_down(sz, q, VOX_CG); si += __shfl_down(si, q, VOX_CG);
        }
        if (act && sub == 0) {
            lvi_pt* out = a.concat ? a.st[0].out : a.st[s].out;
            out[g.out_off + v] = fx_centroid(g, vkey, sx, sy, sz, si, (unsigned)(e - b));
        }
    }
}

// =====================================================================================================
// BINNED path.  bin = voxel idx >> bin_shift; with a compact grid (div_b product <= VB_NB << VB_CL_LOG) a bin
// is 1024 consecutive voxel indices and fits LDS; larger grids get wider bins that are swept once per
// occupied 1024-voxel sub-range.  Nothing is sorted and nothing needs a stable order: integer sums commute.
// =====================================================================================================
// Runs of equal consecutive values across the lanes of a wave (consecutive points of a cloud mostly share a
// bin / a voxel, and same-address LDS atomics serialise).  head = first lane of its run; hpos = that lane;
// len (valid on head lanes) = lanes in the run.
struct WaveRun { bool head; int hpos; int len; };
__device__ __forceinline__ WaveRun wave_runs(unsigned v)
{
    const int l = lane_id();
    const unsigned prev = __shfl_up(v, 1, 64);
    WaveRun r;
    r.head = l == 0 || prev != v;
    const uint64_t hm = __ballot(r.head);
    r.hpos = 63 - __clzll(hm & ((2ull << l) - 1ull));
    const uint64_t above = l == 63 ? 0ull : hm & ~((2ull << l) - 1ull);
    r.len = (above ? __ffsll((unsigned long long)above) - 1 : 64) - l;
    return r;
}

__global__ __launch_bounds__(256) void vb_hist_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y;
    const int n = a.d_n[s];
    const VoxGrid& g = a.grid[s];
    if ((int)blockIdx.x * VB_TILE >= n || g.nbins == 0) return;
    __shared__ unsigned cnt[VB_NB];
    const int nbins = g.nbins, sh = g.bin_shift;
    for (int b = threadIdx.x; b < nbins; b += 256) cnt[b] = 0u;
    __syncthreads();
    const int off = a.dyn[s].in_off;
    const lvi_pt* __restrict__ in = a.st[s].in + off;
    const uint8_t* __restrict__ mask = a.st[s].mask ? a.st[s].mask + off : nullptr;
    // grid-stride over the tiles: one LDS histogram and one flush per workgroup, however many tiles it takes
    const VoxKeyK kk = vox_keyk_of(g);
    for (int base = blockIdx.x * VB_TILE; base < n; base += gridDim.x * VB_TILE) {
        for (int u0 = 0; u0 < VB_TILE / 256; u0 += 8) {                  // eight unconditional loads in flight (clamped index, masked lanes)
            lvi_pt p[8]; bool keep[8];
            vox_load_pts<8, 256>(in, mask, base + u0 * 256 + threadIdx.x, n, p, keep);
#pragma unroll
            for (int u = 0; u < 8; u++) if (keep[u]) atomicAdd(&cnt[vox_key_k(kk, p[u]) >> sh], 1u);
        }
    }
    __syncthreads();
    unsigned* gc = a.binCount + (size_t)s * VB_NB * VB_PAD;
    for (int b = threadIdx.x; b < nbins; b += 256) { const unsigned c = cnt[b]; if (c) atomicAdd(&gc[(size_t)b * VB_PAD], c); }
}

// ---- deterministic partition (plans that cache: the raw local map) ------------------------------------------------------
// The reservation scheme above lets ~1 200 workgroups append 21-point pieces to the advancing ends of the bins: nearly every
// 128-byte line of the bucketed array is shared between workgroups, which is what the 55 us of vb_scatter/map are (DESIGN §9
// of round 1).  Here workgroup w owns the contiguous point range [w K, (w+1) K) (K = a few 4 096-point tiles, VB_WG
// workgroups at most), the per-(workgroup, bin) counts are taken once where the map is written (vb_hist_w + vb_colscan:
// they depend on the input alone), and the scatter places workgroup w's points of bin b at binStart[b] + prefix[w][b]:
// every workgroup writes ONE contiguous piece per bin (~80 points), filled tile after tile by the same workgroup — lines are
// shared only at the piece ends — with no global atomics at all.
__device__ __forceinline__ int vb_wg_points(int n) { const int k = (n + VB_STILE * VB_WG - 1) / (VB_STILE * VB_WG); return VB_STILE * max(k, 1); }

// The reference runs getMinMax3D and then bins every point, for every scan (VoxelGrid::applyFilter, mapOptimization.cpp:958-965).
// Here both happen in ONE pass over the raw map: workgroup w takes the bbox partial record of its contiguous range and, in the
// same sweep, the per-bin counts of that range under the grid geometry the PREVIOUS run of this plan left in a.grid[s] — the
// same geometry whenever the map's extent did not move by a voxel, which vox_setup verifies against the bbox found here
// (g.plan_ok); a point outside that grid, or a changed grid, sends the counts through vb_hist_w once more.
__global__ __launch_bounds__(256) void vb_plan_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y, w = blockIdx.x;
    const int n = seg_len(a, s);
    if (w == 0 && threadIdx.x == 0) { a.d_n[s] = n; }
    const VoxGrid& g = a.grid[s];
    __shared__ unsigned cnt[VB_NB];
    __shared__ int smiss;
    const bool spec = g.n_valid > 0 && !g.overflow && g.nbins > 0 && g.nbins <= VB_NB;
    const int nbins = spec ? g.nbins : 0, sh = g.bin_shift;
    const int d0 = g.div_b[0], d1 = g.div_b[1], d2 = g.div_b[2];
    for (int b = threadIdx.x; b < nbins; b += 256) cnt[b] = 0u;
    if (threadIdx.x == 0) smiss = spec ? 0 : 1;
    __syncthreads();
    const int K = vb_wg_points(n);
    const int i0 = min(n, w * K), i1 = min(n, i0 + K);
    const int off = a.dyn[s].in_off;
    const lvi_pt* __restrict__ in = a.st[s].in + off;
    const uint8_t* __restrict__ mask = a.st[s].mask ? a.st[s].mask + off : nullptr;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float imn = INFINITY, imx = -INFINITY;
    int c = 0, miss = 0;
    const VoxKeyK kk = vox_keyk_of(g);
    for (int base = i0; base < i1; base += VB_STILE) {
        for (int u0 = 0; u0 < VB_STILE / 256; u0 += 8) {
            lvi_pt p[8]; bool keep[8];
            vox_load_pts<8, 256>(in, mask, base + u0 * 256 + threadIdx.x, i1, p, keep);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (!keep[u]) continue;
                mn[0] = fminf(mn[0], p[u].x); mn[1] = fminf(mn[1], p[u].y); mn[2] = fminf(mn[2], p[u].z);
                mx[0] = fmaxf(mx[0], p[u].x); mx[1] = fmaxf(mx[1], p[u].y); mx[2] = fmaxf(mx[2], p[u].z);
                imn = fminf(imn, p[u].intensity); imx = fmaxf(imx, p[u].intensity);
                c++;
                if (spec) {
                    const int ijk0 = (int)sub_rn(floorf(mul_rn(p[u].x, kk.inv)), kk.mb0);
                    const int ijk1 = (int)sub_rn(floorf(mul_rn(p[u].y, kk.inv)), kk.mb1);
                    const int ijk2 = (int)sub_rn(floorf(mul_rn(p[u].z, kk.inv)), kk.mb2);
                    if ((unsigned)ijk0 < (unsigned)d0 && (unsigned)ijk1 < (unsigned)d1 && (unsigned)ijk2 < (unsigned)d2)
                        atomicAdd(&cnt[((unsigned)ijk0 + (unsigned)ijk1 * kk.mul1 + (unsigned)ijk2 * kk.mul2) >> sh], 1u);
                    else miss = 1;
                }
            }
        }
    }
    // bbox partial record of this workgroup (vox_setup folds the records; those of ranges beyond the input stay empty)
    __shared__ float smn[4][4], smx[4][4];
    __shared__ int scnt[4];
    c = wave_sum(c);
#pragma unroll
    for (int d = 0; d < 3; d++) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    imn = wave_min(imn); imx = wave_max(imx);
    if (lane_id() == 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) { smn[wave_id()][d] = mn[d]; smx[wave_id()][d] = mx[d]; }
        smn[wave_id()][3] = imn; smx[wave_id()][3] = imx;
        scnt[wave_id()] = c;
    }
    if (miss) smiss = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int cc = 0;
        float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int q = 0; q < 4; q++) {
            cc += scnt[q];
#pragma unroll
            for (int d = 0; d < 4; d++) { lo[d] = fminf(lo[d], smn[q][d]); hi[d] = fmaxf(hi[d], smx[q][d]); }
        }
        if (w < a.nblk_mm) {                        // (a range that holds points always has w < nblk_mm)
            float* rec = a.mmPartial + ((size_t)s * a.nblk_mm + w) * 12;
            rec[0] = lo[0]; rec[1] = lo[1]; rec[2] = lo[2]; rec[3] = hi[0]; rec[4] = hi[1]; rec[5] = hi[2];
            rec[6] = __int_as_float(cc); rec[7] = lo[3]; rec[8] = hi[3];
        }
        if (w + VB_WG < a.nblk_mm) (a.mmPartial + ((size_t)s * a.nblk_mm + w + VB_WG) * 12)[6] = __int_as_float(0);   // records another realisation's bbox pass may have left
        if (smiss && i0 < i1) atomicOr(&a.planMiss[s], 1);
        if (!spec && w == 0) atomicOr(&a.planMiss[s], 1);
    }
    unsigned* row = a.wprefix + ((size_t)s * VB_WG + w) * VB_WROW;
    if (i0 < i1) for (int b = threadIdx.x; b < nbins; b += 256) row[b] = cnt[b];       // (rows of empty ranges are never read)
}

__global__ __launch_bounds__(256) void vb_hist_w_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y, w = blockIdx.x;
    const int n = a.d_n[s];
    const VoxGrid& g = a.grid[s];
    if (a.plan_spec && g.plan_ok) return;           // vb_plan's counts stand
    const int K = vb_wg_points(n);
    const int i0 = w * K, i1 = min(n, i0 + K);
    if (i0 >= i1) return;                           // (rows of empty ranges are never read)
    __shared__ unsigned cnt[VB_NB];
    const int nbins = g.nbins, sh = g.bin_shift;
    for (int b = threadIdx.x; b < nbins; b += 256) cnt[b] = 0u;
    __syncthreads();
    const int off = a.dyn[s].in_off;
    const lvi_pt* __restrict__ in = a.st[s].in + off;
    const uint8_t* __restrict__ mask = a.st[s].mask ? a.st[s].mask + off : nullptr;
    const VoxKeyK kk = vox_keyk_of(g);
    for (int base = i0; base < i1; base += VB_STILE) {
        for (int u0 = 0; u0 < VB_STILE / 256; u0 += 8) {
            lvi_pt p[8]; bool keep[8];
            vox_load_pts<8, 256>(in, mask, base + u0 * 256 + threadIdx.x, i1, p, keep);
#pragma unroll
            for (int u = 0; u < 8; u++) if (keep[u]) atomicAdd(&cnt[vox_key_k(kk, p[u]) >> sh], 1u);
        }
    }
    __syncthreads();
    unsigned* row = a.wprefix + ((size_t)s * VB_WG + w) * VB_WROW;
    for (int b = threadIdx.x; b < nbins; b += 256) row[b] = cnt[b];
}

// per bin: exclusive prefix over the workgroups (in place) and the bin's total.  16 bins x 16 parts of the workgroup range per
// workgroup of this kernel: a part's column piece (32 rows) is loaded at once, summed, the parts meet in LDS, then each writes its
// prefixes from the values it still holds (one column walked by ONE thread was 512 dependent steps: 65 us for eight slots; four
// quarters with a second read of the column: 40 us; only the workgroups of live bins do anything)
__global__ __launch_bounds__(256) void vb_colscan_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    unsigned* totals = a.binCountOut;
    const int s = blockIdx.y;
    constexpr int NP = 16, PW = VB_WG / NP;                              // parts, rows per part
    static_assert(PW == 32, "a part's rows are held in 32 registers");
    const int b = blockIdx.x * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
    __shared__ unsigned psum[NP][16];
    const bool live = b < VB_NB && b < a.grid[s].nbins;
    unsigned* col = a.wprefix + (size_t)s * VB_WG * VB_WROW + (live ? b : 0) + (size_t)part * PW * VB_WROW;
    // rows of workgroups whose point range is empty are neither written (vb_plan, vb_hist_w) nor read (vb_scatter_det): the corner map
    // of the bench — 4 096 bins, fifteen active workgroups — was 8 MB of zeros read and written per slot
    const int n = a.d_n[s];
    const int nrow = min(VB_WG, (n + vb_wg_points(n) - 1) / vb_wg_points(n)) - part * PW;      // active rows of this part
    unsigned v[PW];
    unsigned acc = 0u;
#pragma unroll
    for (int u = 0; u < PW; u++) v[u] = (live && u < nrow) ? col[(size_t)u * VB_WROW] : 0u;
#pragma unroll
    for (int u = 0; u < PW; u++) acc += v[u];
    psum[part][threadIdx.x & 15] = acc;
    __syncthreads();
    unsigned before = 0u, total = 0u;
#pragma unroll
    for (int q = 0; q < NP; q++) { const unsigned t = psum[q][threadIdx.x & 15]; before += q < part ? t : 0u; total += t; }
    if (live) {
        unsigned run = before;
#pragma unroll
        for (int u = 0; u < PW; u++) { if (u < nrow) col[(size_t)u * VB_WROW] = run; run += v[u]; }
    }
    if (part == 0 && b < VB_NB) totals[(size_t)s * VB_NB + b] = live ? total : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.plan_spec) a.planMiss[s] = 0;      // consumed (vox_setup read it): the next run's vb_plan starts clean
}

__global__ __launch_bounds__(256) void vb_scatter_det_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y, w = blockIdx.x;
    const int n = a.d_n[s];
    const VoxGrid& g = a.grid[s];
    const int K = vb_wg_points(n);
    const int i0 = w * K, i1 = min(n, i0 + K);
    if (i0 >= i1 || g.nbins == 0) return;
    __shared__ unsigned cnt[VB_NB], pos[VB_NB];
    const int nbins = g.nbins, sh = g.bin_shift;
    const int* bs = a.binStart + (size_t)s * (VB_NB + 1);
    const unsigned* row = a.wprefix + ((size_t)s * VB_WG + w) * VB_WROW;
    for (int b = threadIdx.x; b < nbins; b += 256) { cnt[b] = 0u; pos[b] = (unsigned)bs[b] + row[b]; }
    __syncthreads();
    const int off = a.dyn[s].in_off;
    const lvi_pt* __restrict__ in = a.st[s].in + off;
    const uint8_t* __restrict__ mask = a.st[s].mask ? a.st[s].mask + off : nullptr;
    lvi_pt* __restrict__ dst = a.bucketed + (size_t)s * a.seg_cap;
    constexpr int IT = VB_STILE / 256;
    const VoxKeyK kk = vox_keyk_of(g);
    for (int base = i0; base < i1; base += VB_STILE) {
        lvi_pt p[IT]; int bin[IT]; unsigned rk[IT]; bool keep[IT];
        // unconditional loads (index clamped into the range, the lane masked afterwards): behind a per-element branch every
        // load of the tile waited for the one before it
        vox_load_pts<IT, 256>(in, mask, base + threadIdx.x, i1, p, keep);
#pragma unroll
        for (int u = 0; u < IT; u++) bin[u] = keep[u] ? (int)(vox_key_k(kk, p[u]) >> sh) : -1;
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const WaveRun r = wave_runs((unsigned)bin[u]);
            unsigned b0 = 0u;
            if (bin[u] >= 0 && r.head) b0 = atomicAdd(&cnt[bin[u]], (unsigned)r.len);
            rk[u] = __shfl(b0, r.hpos, 64) + (unsigned)(lane_id() - r.hpos);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < IT; u++) if (bin[u] >= 0) dst[pos[bin[u]] + rk[u]] = p[u];
        __syncthreads();
        for (int b = threadIdx.x; b < nbins; b += 256) { pos[b] += cnt[b]; cnt[b] = 0u; }
        __syncthreads();
    }
}

// per segment: binStart = exclusive scan of binCount, cursor = binStart, binCount back to zero
__global__ __launch_bounds__(256) void vb_scan_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.x;
    const int nbins = a.grid[s].nbins;
    unsigned* gc = a.binCount + (size_t)s * VB_NB * VB_PAD;
    int* bs = a.binStart + (size_t)s * (VB_NB + 1);
    unsigned* cur = a.cursor + (size_t)s * VB_NB * VB_PAD;
    __shared__ int ws[8];
    constexpr int PER = VB_NB / 256;                    // consecutive bins per thread
    // (one source pointer chosen up front: a per-element choice between two arrays serialised the 16 loads of a thread)
    const unsigned* cnt_src = a.binCountCached ? a.binCountCached + (size_t)s * VB_NB : gc;
    const size_t cnt_stride = a.binCountCached ? 1 : VB_PAD;
    int v[PER], sum = 0, csum = 0, msum = 0, lsum = 0;
    const bool wide = a.grid[s].bin_shift > VB_CL_LOG;
    // chunks of a bin: none for a light bin (<= VB_LIGHT points: one wavefront of vb_light_kernel takes it), else ceil(n / VB_CH)
    auto chunks_of = [&](int n) { return wide ? (n > 0 ? 1 : 0) : (n <= VB_LIGHT ? 0 : (n + a.ch - 1) / a.ch); };
#pragma unroll
    for (int j = 0; j < PER; j++) v[j] = (int)cnt_src[(size_t)min((int)threadIdx.x * PER + j, VB_NB - 1) * cnt_stride];   // all 16 loads in flight
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int b = threadIdx.x * PER + j;
        v[j] = b < nbins ? v[j] : 0;
        sum += v[j];
        const int nch = chunks_of(v[j]);
        csum += nch; msum += nch > 1 ? nch : 0;
        lsum += (!wide && v[j] > 0 && v[j] <= VB_LIGHT) ? 1 : 0;
    }
    int tot, ctot, mtot, ltot;
    int ex, cex, mex, lex;
    // the four prefix sums in ONE scan of a packed word (points 24 bits, chunks 14, chunks of multi-chunk bins 13, light bins 13) when the
    // plan's capacities fit the fields — every plan of the library does; four scans were twelve barriers of this one-workgroup kernel
    if (a.seg_cap < (1 << 24) && a.max_chunks < (1 << 14) && a.max_multi < (1 << 13)) {
        __shared__ unsigned long long ws64[8];
        unsigned long long t64;
        const unsigned long long e64 = block_excl_scan_u64<256>((unsigned long long)sum | ((unsigned long long)csum << 24) | ((unsigned long long)msum << 38) |
                                                                    ((unsigned long long)lsum << 51), ws64, &t64);
        ex = (int)(e64 & 0xFFFFFFull); cex = (int)((e64 >> 24) & 0x3FFFull); mex = (int)((e64 >> 38) & 0x1FFFull); lex = (int)(e64 >> 51);
        tot = (int)(t64 & 0xFFFFFFull); ctot = (int)((t64 >> 24) & 0x3FFFull); mtot = (int)((t64 >> 38) & 0x1FFFull); ltot = (int)(t64 >> 51);
    } else {
        ex = block_excl_scan<256>(sum, ws, &tot);
        cex = block_excl_scan<256>(csum, ws, &ctot);
        mex = block_excl_scan<256>(msum, ws, &mtot);
        lex = block_excl_scan<256>(lsum, ws, &ltot);
    }
    int* cs = a.chunkStart + (size_t)s * (VB_NB + 1);
    int* ms = a.multiStart + (size_t)s * (VB_NB + 1);
    int* lb = a.lightBin + (size_t)s * (VB_NB + 1);
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int b = threadIdx.x * PER + j;
        if (b < nbins) { bs[b] = ex; cur[(size_t)b * VB_PAD] = (unsigned)ex; if (!a.binCountCached) gc[(size_t)b * VB_PAD] = 0u; cs[b] = cex; ms[b] = mex; a.binVox[(size_t)s * VB_NB + b] = 0; }
        ex += v[j];
        const int nch = chunks_of(v[j]);
        for (int q = 0; q < nch; q++) a.chunkBin[(size_t)s * a.max_chunks + cex + q] = b;
        if (nch > 1) for (int q = 0; q < nch; q++) a.multiOwner[(size_t)s * a.max_multi + mex + q] = q == 0 ? b : -1;
        cex += nch; mex += nch > 1 ? nch : 0;
        if (!wide && v[j] > 0 && v[j] <= VB_LIGHT) lb[lex++] = b;
    }
    if (threadIdx.x == 0) { bs[nbins] = tot; cs[nbins] = ctot; ms[nbins] = mtot; lb[VB_NB] = ltot; }
}

__global__ __launch_bounds__(256) void vb_scatter_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y;
    const int n = a.d_n[s];
    const int base = blockIdx.x * VB_STILE;
    const VoxGrid& g = a.grid[s];
    if (base >= n || g.nbins == 0) return;
    __shared__ unsigned cnt[VB_NB];
    const int nbins = g.nbins, sh = g.bin_shift;
    for (int b = threadIdx.x; b < nbins; b += 256) cnt[b] = 0u;
    __syncthreads();
    const int off = a.dyn[s].in_off;
    const lvi_pt* __restrict__ in = a.st[s].in + off;
    const uint8_t* __restrict__ mask = a.st[s].mask ? a.st[s].mask + off : nullptr;
    constexpr int IT = VB_STILE / 256;
    lvi_pt p[IT]; int bin[IT]; unsigned rk[IT]; bool keep[IT];
    const VoxKeyK kk = vox_keyk_of(g);
    vox_load_pts<IT, 256>(in, mask, base + threadIdx.x, n, p, keep);     // unconditional loads, clamped index (base < n here), lanes masked afterwards
#pragma unroll
    for (int u = 0; u < IT; u++) bin[u] = keep[u] ? (int)(vox_key_k(kk, p[u]) >> sh) : -1;
#pragma unroll
    for (int u = 0; u < IT; u++) {                      // rank inside (tile, bin): any order will do; one LDS atomic per run of lanes,
        const WaveRun r = wave_runs((unsigned)bin[u]);  // so that a run lands on consecutive addresses (measured: 56 vs 59 us)
        unsigned b0 = 0u;
        if (bin[u] >= 0 && r.head) b0 = atomicAdd(&cnt[bin[u]], (unsigned)r.len);
        rk[u] = __shfl(b0, r.hpos, 64) + (unsigned)(lane_id() - r.hpos);
    }
    __syncthreads();
    unsigned* cur = a.cursor + (size_t)s * VB_NB * VB_PAD;
    {   // one global reservation per occupied (tile, bin); all of a thread's atomics are issued before the first result is used
        constexpr int PER = VB_NB / 256;
        unsigned c[PER], gb[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) { const int b = threadIdx.x + j * 256; c[j] = b < nbins ? cnt[b] : 0u; }
#pragma unroll
        for (int j = 0; j < PER; j++) { gb[j] = 0u; if (c[j]) gb[j] = atomicAdd(&cur[(size_t)(threadIdx.x + j * 256) * VB_PAD], c[j]); }
#pragma unroll
        for (int j = 0; j < PER; j++) if (c[j]) cnt[threadIdx.x + j * 256] = gb[j];
    }
    __syncthreads();
    lvi_pt* __restrict__ dst = a.bucketed + (size_t)s * a.seg_cap;
#pragma unroll
    for (int u = 0; u < IT; u++) if (bin[u] >= 0) dst[cnt[bin[u]] + rk[u]] = p[u];
}

// LDS accumulators of one sweep: up to 1024 consecutive voxels
struct VbCells {
    unsigned long long sx[1 << VB_CL_LOG], sy[1 << VB_CL_LOG], sz[1 << VB_CL_LOG], si[1 << VB_CL_LOG];
    unsigned cn[1 << VB_CL_LOG];
};

template <int NT = 256>
__device__ __forceinline__ void vb_zero(VbCells& L, int cells)
{
    for (int c = threadIdx.x; c < cells; c += NT) { L.sx[c] = 0ull; L.sy[c] = 0ull; L.sz[c] = 0ull; L.si[c] = 0ull; L.cn[c] = 0u; }
    __syncthreads();
}

// add bucketed points [q0, q1) whose voxel idx lies in [k0, k0 + cells) to the LDS accumulators
template <int NT = 256>
__device__ __forceinline__ void vb_add_points(VbCells& L, const VoxGrid& g, const lvi_pt* __restrict__ pts, int q0, int q1, unsigned k0, int cells)
{
    constexpr int NL = 8;                               // loads in flight per lane
    const VoxFx f = vox_fx_of(g);
    for (int i0 = q0 + threadIdx.x; i0 < q1; i0 += NL * NT) {
        lvi_pt p[NL]; bool ok[NL];
#pragma unroll
        for (int u = 0; u < NL; u++) { const int i = i0 + u * NT; ok[u] = i < q1; p[u] = pts[min(i, q1 - 1)]; }    // unconditional loads, lanes masked below
#pragma unroll
        for (int u = 0; u < NL; u++) {
            unsigned long long v[4];
            const unsigned c = vox_fx_point(f, p[u], v) - k0;
            if (!ok[u] || c >= (unsigned)cells) continue;               // (c: another sub-range of a wide bin)
            atomicAdd(&L.sx[c], v[0]); atomicAdd(&L.sy[c], v[1]); atomicAdd(&L.sz[c], v[2]); atomicAdd(&L.si[c], v[3]);
            atomicAdd(&L.cn[c], 1u);
        }
    }
    __syncthreads();
}

// ordered compaction of the occupied voxels of the sweep into the staging area (thread t owns voxels 4t..4t+3);
// returns the number written
template <int NT = 256>
__device__ __forceinline__ int vb_emit(const VbCells& L, const VoxGrid& g, lvi_pt* __restrict__ stg, uint2* __restrict__ skc, int at, unsigned k0, int cells, int* ws,
                                       unsigned short* cl)
{
    // the occupied cells, in idx order, first as a list in LDS (thread t scans cells PER t .. PER t + PER - 1), then one
    // centroid per thread: a bin of a dense map has ~90 occupied cells, which 90 threads finish in one step instead of
    // ~25 threads in four
    constexpr int PER = (1 << VB_CL_LOG) / NT;
    const int tid = threadIdx.x;
    int c4 = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { const int c = tid * PER + j; c4 += c < cells && L.cn[c] != 0u; }
    int tot;
    int r = block_excl_scan<NT>(c4, ws, &tot);
#pragma unroll
    for (int j = 0; j < PER; j++) { const int c = tid * PER + j; if (c < cells && L.cn[c] != 0u) cl[r++] = (unsigned short)c; }
    __syncthreads();
    for (int q = tid; q < tot; q += NT) {
        const int c = cl[q];
        const unsigned m = L.cn[c];
        stg[at + q] = fx_centroid(g, k0 + (unsigned)c, L.sx[c], L.sy[c], L.sz[c], L.si[c], m); skc[at + q] = make_uint2(k0 + (unsigned)c, m);
    }
    __syncthreads();
    return tot;
}

// Light bins (<= VB_LIGHT points; the ring and scan grids consist of them, and so does the corner map): ONE wavefront per bin,
// four bins in flight per workgroup, no workgroup barrier.  The occupied cells of the bin are a 1024-bit set; a cell's rank in
// that set is its slot in a compact table (<= VB_LIGHT slots) and its position in the output, so nothing is zeroed, scanned
// or compacted beyond the cells that hold points (a 1024-cell table per 40-point bin was 34 us for a 100 k-point ring grid).
struct VbLight {
    unsigned long long bm[16]; unsigned base[16];
    unsigned long long sx[VB_LIGHT], sy[VB_LIGHT], sz[VB_LIGHT], si[VB_LIGHT];
    unsigned cn[VB_LIGHT]; unsigned short cell[VB_LIGHT];
};
// LDS written by some lanes of a wavefront is read by others: order the accesses (a wavefront's LDS operations complete in order)
__device__ __forceinline__ void wave_lds_sync() { __threadfence_block(); __builtin_amdgcn_wave_barrier(); }

// (the second role of the workgroups of vb_accum_kernel: called by all 256 threads, after a workgroup barrier)
__device__ __forceinline__ void vb_light_items(const VoxArgs& a, int s, const VoxGrid& g, VbLight* LW)
{
    const int sh = g.bin_shift;
    if (sh > VB_CL_LOG) return;
    const int* lb = a.lightBin + (size_t)s * (VB_NB + 1);
    const int nl = lb[VB_NB];
    const int wv = threadIdx.x >> 6, ln = lane_id();
    VbLight& L = LW[wv];
    const int* bs = a.binStart + (size_t)s * (VB_NB + 1);
    const lvi_pt* __restrict__ pts = a.bucketed + (size_t)s * a.seg_cap;
    lvi_pt* __restrict__ stg = a.staging + (size_t)s * a.seg_cap;
    uint2* __restrict__ skc = a.stagingKC + (size_t)s * a.seg_cap;
    constexpr int PP = VB_LIGHT / 64;                   // points per lane
    const VoxFx f = vox_fx_of(g);
    for (int item = blockIdx.x * 4 + wv; item < nl; item += gridDim.x * 4) {
        const int b = lb[item];
        const int p0 = bs[b], p1 = bs[b + 1];
        const unsigned kbase = (unsigned)b << sh;
        if (ln < 16) L.bm[ln] = 0ull;
        wave_lds_sync();
        lvi_pt p[PP]; unsigned c[PP]; bool ok[PP]; unsigned long long v[PP][4];
#pragma unroll
        for (int u = 0; u < PP; u++) { const int i = p0 + ln + 64 * u; ok[u] = i < p1; p[u] = pts[min(i, p1 - 1)]; }      // p0 < p1 for a listed bin
#pragma unroll
        for (int u = 0; u < PP; u++) {
            c[u] = (vox_fx_point(f, p[u], v[u]) - kbase) & (unsigned)(VB_TAB - 1);
            if (ok[u]) atomicOr(&L.bm[c[u] >> 6], 1ull << (c[u] & 63u));
        }
        wave_lds_sync();
        const int pc = ln < 16 ? __popcll(L.bm[ln]) : 0;
        const int incl = wave_incl_scan(pc);
        if (ln < 16) L.base[ln] = (unsigned)(incl - pc);
        const int nd = __shfl(incl, 15, 64);            // occupied cells of the bin (<= its points <= VB_LIGHT)
        for (int r = ln; r < nd; r += 64) { L.sx[r] = 0ull; L.sy[r] = 0ull; L.sz[r] = 0ull; L.si[r] = 0ull; L.cn[r] = 0u; }
        wave_lds_sync();
#pragma unroll
        for (int u = 0; u < PP; u++) {
            if (!ok[u]) continue;
            const unsigned w = c[u] >> 6;
            const int slot = (int)L.base[w] + __popcll(L.bm[w] & ((1ull << (c[u] & 63u)) - 1ull));
            atomicAdd(&L.sx[slot], v[u][0]); atomicAdd(&L.sy[slot], v[u][1]); atomicAdd(&L.sz[slot], v[u][2]); atomicAdd(&L.si[slot], v[u][3]);
            atomicAdd(&L.cn[slot], 1u);
            L.cell[slot] = (unsigned short)c[u];        // every point of the cell writes the same value
        }
        wave_lds_sync();
        for (int r = ln; r < nd; r += 64) {
            const unsigned key = kbase + (unsigned)L.cell[r];
            const unsigned m = L.cn[r];
            stg[p0 + r] = fx_centroid(g, key, L.sx[r], L.sy[r], L.sz[r], L.si[r], m); skc[p0 + r] = make_uint2(key, m);
        }
        if (ln == 0) a.binVox[(size_t)s * VB_NB + b] = nd;
        wave_lds_sync();
    }
}

// One workgroup per chunk of <= VB_CH bucketed points of one bin.  A bin that is a single chunk is finished here;
// the chunks of a larger bin leave their LDS tables in chunkTab and vb_merge adds them up (no global atomics,
// so a bin with 10^5 points is spread over 25 workgroups instead of keeping one busy for 0.2 ms).
template <int NT>
__global__ __launch_bounds__(NT) void vb_accum_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y;
    const VoxGrid& g = a.grid[s];
    const int nbins = g.nbins, sh = g.bin_shift;
    if (nbins == 0) return;
    union Lds { struct { VbCells L; unsigned short cl[1 << VB_CL_LOG]; } c; VbLight w[4]; };
    __shared__ Lds U;                                                       // chunk role: table + vb_emit's list of occupied cells; light role: 4 wavefront tables
    VbCells& L = U.c.L;
    unsigned short* cl = U.c.cl;
    __shared__ unsigned occ[32];                                            // occupied sub-ranges of a wide bin, one bit each
    static_assert(VB_NB == 4096 && VB_CL_LOG == 10, "occ[] is sized for 2^32 / VB_NB / 2^VB_CL_LOG sub-ranges");
    __shared__ int ws[NT / 64 + 1];
    const int* bs = a.binStart + (size_t)s * (VB_NB + 1);
    const int* cs = a.chunkStart + (size_t)s * (VB_NB + 1);
    const int* ms = a.multiStart + (size_t)s * (VB_NB + 1);
    const lvi_pt* __restrict__ pts = a.bucketed + (size_t)s * a.seg_cap;
    lvi_pt* __restrict__ stg = a.staging + (size_t)s * a.seg_cap;
    uint2* __restrict__ skc = a.stagingKC + (size_t)s * a.seg_cap;
    constexpr int PER = (1 << VB_CL_LOG) / NT;
    const int tid = threadIdx.x;
    const int nchunks = cs[nbins];
    const int cells = sh >= VB_CL_LOG ? (1 << VB_CL_LOG) : (1 << sh);
    for (int w = blockIdx.x; w < nchunks; w += gridDim.x) {
        const int b = a.chunkBin[(size_t)s * a.max_chunks + w];
        const int j = w - cs[b], nch = cs[b + 1] - cs[b];
        const int p0 = bs[b], p1 = bs[b + 1];
        const unsigned kbase = (unsigned)b << sh;
        if (sh > VB_CL_LOG) {
            // wide bin (sparse grid): one workgroup, one sweep per occupied 1024-voxel sub-range
            const int nsub = 1 << (sh - VB_CL_LOG);
            if (tid < 32) occ[tid] = 0u;
            __syncthreads();
            for (int i = p0 + tid; i < p1; i += NT) { const unsigned q = (vox_key_of_pt(g, pts[i]) - kbase) >> VB_CL_LOG; atomicOr(&occ[q >> 5], 1u << (q & 31)); }
            __syncthreads();
            int carry = 0;
            for (int sp = 0; sp < nsub; sp++) {
                if (!((occ[sp >> 5] >> (sp & 31)) & 1u)) continue;             // same for every thread
                const unsigned k0 = kbase + ((unsigned)sp << VB_CL_LOG);
                vb_zero<NT>(L, cells);
                vb_add_points<NT>(L, g, pts, p0, p1, k0, cells);
                carry += vb_emit<NT>(L, g, stg, skc, p0 + carry, k0, cells, ws, cl);
            }
            if (tid == 0) a.binVox[(size_t)s * VB_NB + b] = carry;
            continue;
        }
        const int q0 = p0 + j * a.ch, q1 = min(p1, q0 + a.ch);
        vb_zero<NT>(L, cells);
        vb_add_points<NT>(L, g, pts, q0, q1, kbase, cells);
        if (nch == 1) {
            const int tot = vb_emit<NT>(L, g, stg, skc, p0, kbase, cells, ws, cl);
            if (tid == 0) a.binVox[(size_t)s * VB_NB + b] = tot;
        } else {
            // leave the OCCUPIED cells of the table in global memory for vb_merge, compacted (a bin of a dense map holds ~90
            // voxels in its 1024 cells: 4 KB instead of the 36 KB of the whole table, written and read once per chunk).
            // (Letting the workgroup that finishes a bin's last chunk add them up needs a device-scope fence per chunk; on a
            // multi-XCD part every such fence writes back and invalidates the XCD's L2 — measured 257 us instead of 41 + 19 us
            // for the 4.9M-point map.)
            unsigned long long* tv = a.chunkTabV + ((size_t)s * a.max_multi + ms[b] + j) * (size_t)(4 * VB_TAB);
            unsigned* tc = a.chunkTabC + ((size_t)s * a.max_multi + ms[b] + j) * (size_t)VB_TABC;
            int c4 = 0;
#pragma unroll
            for (int q = 0; q < PER; q++) { const int c = tid * PER + q; c4 += c < cells && L.cn[c] != 0u; }
            int tot;
            int r = block_excl_scan<NT>(c4, ws, &tot);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int c = tid * PER + q;
                const unsigned m = c < cells ? L.cn[c] : 0u;                 // m <= a.ch < 2^16, c < 2^10
                if (m) { tv[r] = L.sx[c]; tv[VB_TAB + r] = L.sy[c]; tv[2 * VB_TAB + r] = L.sz[c]; tv[3 * VB_TAB + r] = L.si[c]; tc[r] = ((unsigned)c << 16) | m; r++; }
            }
            if (tid == 0) tc[VB_TAB] = (unsigned)tot;
            __syncthreads();
        }
    }
    // second role: the light bins of the segment, one wavefront each
    __syncthreads();
    if constexpr (NT == 256) vb_light_items(a, s, g, U.w);
}

// bins of more than one chunk: add the chunk tables up and emit
__global__ __launch_bounds__(256) void vb_merge_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y;
    const VoxGrid& g = a.grid[s];
    const int nbins = g.nbins, sh = g.bin_shift;
    if (nbins == 0 || sh > VB_CL_LOG) return;
    __shared__ VbCells L;
    __shared__ int ws[8];
    __shared__ unsigned short cl[1 << VB_CL_LOG];
    const int* bs = a.binStart + (size_t)s * (VB_NB + 1);
    const int* cs = a.chunkStart + (size_t)s * (VB_NB + 1);
    const int* ms = a.multiStart + (size_t)s * (VB_NB + 1);
    const int mtot = ms[nbins];                                             // chunk tables of this segment: none for the ring / scan grids
    lvi_pt* __restrict__ stg = a.staging + (size_t)s * a.seg_cap;
    uint2* __restrict__ skc = a.stagingKC + (size_t)s * a.seg_cap;
    const int cells = 1 << sh;
    // one workgroup per multi-chunk bin, found through its first chunk-table slot (at most n / VB_CH such bins exist: the grid
    // is sized for that, not for the bins — every workgroup of this kernel needs 39 KB of LDS just to start)
    for (int t = blockIdx.x; t < mtot; t += gridDim.x) {
        const int b = a.multiOwner[(size_t)s * a.max_multi + t];
        if (b < 0) continue;
        const int nch = cs[b + 1] - cs[b];
        const size_t t0 = (size_t)s * a.max_multi + ms[b];
        vb_zero(L, cells);
        // wavefront w adds chunks w, w + 4, ...: entries of one chunk name distinct cells, chunks meet through LDS atomics;
        // four entries per lane in flight, the next chunk's entry count fetched with them
        const int wv = threadIdx.x >> 6, ln = lane_id();
        int ne = wv < nch ? (int)(a.chunkTabC + (t0 + wv) * (size_t)VB_TABC)[VB_TAB] : 0;
        for (int j = wv; j < nch; j += 4) {
            const unsigned long long* tv = a.chunkTabV + (t0 + j) * (size_t)(4 * VB_TAB);
            const unsigned* tc = a.chunkTabC + (t0 + j) * (size_t)VB_TABC;
            const int ne_next = j + 4 < nch ? (int)(tc + 4 * (size_t)VB_TABC)[VB_TAB] : 0;
            for (int r0 = ln; r0 < ne; r0 += 256) {
                unsigned e[4]; unsigned long long x[4], y[4], z[4], w[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = r0 + 64 * u;
                    e[u] = 0u;
                    if (r < ne) { e[u] = tc[r]; x[u] = tv[r]; y[u] = tv[VB_TAB + r]; z[u] = tv[2 * VB_TAB + r]; w[u] = tv[3 * VB_TAB + r]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (!e[u]) continue;                                     // a real entry has a count >= 1
                    const int c = (int)(e[u] >> 16);
                    atomicAdd(&L.sx[c], x[u]); atomicAdd(&L.sy[c], y[u]); atomicAdd(&L.sz[c], z[u]); atomicAdd(&L.si[c], w[u]);
                    atomicAdd(&L.cn[c], e[u] & 0xFFFFu);
                }
            }
            ne = ne_next;
        }
        __syncthreads();
        const int tot = vb_emit(L, g, stg, skc, bs[b], (unsigned)b << sh, cells, ws, cl);
        if (threadIdx.x == 0) a.binVox[(size_t)s * VB_NB + b] = tot;
    }
}

// binOut = exclusive scan of binVox per segment; voxel counts and output offsets of the batch
__global__ __launch_bounds__(256) void vb_outscan_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    __shared__ int ws[8];
    constexpr int PER = VB_NB / 256;
    for (int s = 0; s < a.nseg; s++) {
        const VoxGrid& g = a.grid[s];
        const int nbins = g.nbins;
        const int* bv = a.binVox + (size_t)s * VB_NB;
        int* bo = a.binOut + (size_t)s * VB_NB;
        int v[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) v[j] = bv[min((int)threadIdx.x * PER + j, VB_NB - 1)];            // all 16 loads in flight, masked below
#pragma unroll
        for (int j = 0; j < PER; j++) { const int b = threadIdx.x * PER + j; v[j] = b < nbins ? v[j] : 0; sum += v[j]; }
        int tot;
        int ex = block_excl_scan<256>(sum, ws, &tot);
#pragma unroll
        for (int j = 0; j < PER; j++) { const int b = threadIdx.x * PER + j; if (b < nbins) bo[b] = ex; ex += v[j]; }
        if (threadIdx.x == 0) a.grid[s].nvox = g.overflow ? g.n_valid : tot;
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

__global__ __launch_bounds__(256) void vb_copy_kernel(Batch<VoxArgs> B_)
{
    const VoxArgs& a = B_.a[blockIdx.z];
    const int s = blockIdx.y;
    const VoxGrid& g = a.grid[s];
    lvi_pt* __restrict__ out = (a.concat ? a.st[0].out : a.st[s].out) + g.out_off;
    const int tid = threadIdx.x;
    if (!g.overflow) {
        const int* bs = a.binStart + (size_t)s * (VB_NB + 1);
        const int* bv = a.binVox + (size_t)s * VB_NB;
        const int* bo = a.binOut + (size_t)s * VB_NB;
        const lvi_pt* __restrict__ stg = a.staging + (size_t)s * a.seg_cap;
        for (int b = blockIdx.x; b < g.nbins; b += gridDim.x) {
            const int nv = bv[b], src = bs[b], dst = bo[b];
            for (int j = tid; j < nv; j += 256) out[dst + j] = stg[src + j];
        }
        return;
    }
    // PCL's overflow rule: output = the segment's (unmasked) input points, in input order
    const int n = a.d_n[s];
    const int off = a.dyn[s].in_off;
    const lvi_pt* __restrict__ in = a.st[s].in + off;
    const uint8_t* __restrict__ mask = a.st[s].mask ? a.st[s].mask + off : nullptr;
    const int chunk = (n + (int)gridDim.x - 1) / (int)gridDim.x;
    const int i0 = min(n, (int)blockIdx.x * chunk), i1 = min(n, i0 + chunk);
    if (i0 >= i1) return;
    __shared__ int ws[8];
    int basep = i0;
    if (mask) {
        int c = 0;
        for (int i = tid; i < i0; i += 256) c += mask[i] != 0;
        int tot;
        (void)block_excl_scan<256>(c, ws, &tot);
        basep = tot;
    }
    for (int i = i0; i < i1; i += 256) {
        const int j = i + tid;
        const bool keep = j < i1 && (!mask || mask[j]);
        int tot;
        const int r = block_excl_scan<256>(keep ? 1 : 0, ws, &tot);
        if (keep) out[basep + r] = in[j];
        basep += tot;
    }
}


// =====================================================================================================
// A cloud of at most VOX_TINY points (the node's key-pose grid: a few dozen to a few hundred poses per scan, mapOptimization.cpp:
// 894-929) in ONE workgroup and one launch: bounding box, grid geometry (vox_setup_math), keys, order by (key, input index) by
// counting, heads, fixed-point centroids — the general path's expressions, hence its bits — read from and written to pinned host
// memory.  The general path is nine launches, an upload and two reads for the same cloud: ~100 us of latency per sequential scan.
// =====================================================================================================
struct VoxTinyArgs {
    const lvi_pt* in; int n; float leaf; int seg_cap, bin_pts, bin_max;
    lvi_pt* out; int* hdr;                            // hdr: {voxels, overflow}
    int* cells; int* counts; int* keys;               // the debug views' voxel idx / points per voxel / voxel idx per input point
};
__global__ __launch_bounds__(VOX_TINY) void vox_tiny_kernel(VoxTinyArgs a)
{
    __shared__ lvi_pt spt[VOX_TINY];
    __shared__ unsigned long long sk[VOX_TINY], ssorted[VOX_TINY];
    __shared__ VoxGrid sg;
    __shared__ float smn[VOX_TINY / 64][4], smx[VOX_TINY / 64][4];
    __shared__ int ws[VOX_TINY / 64 + 1];
    const int i = threadIdx.x, n = a.n;
    const bool in_range = i < n;
    lvi_pt p = {0.f, 0.f, 0.f, 0.f};
    if (in_range) p = a.in[i];
    spt[i] = p;
    float lo[4] = {in_range ? p.x : INFINITY, in_range ? p.y : INFINITY, in_range ? p.z : INFINITY, in_range ? p.intensity : INFINITY};
    float hi[4] = {in_range ? p.x : -INFINITY, in_range ? p.y : -INFINITY, in_range ? p.z : -INFINITY, in_range ? p.intensity : -INFINITY};
#pragma unroll
    for (int d = 0; d < 4; d++) { lo[d] = wave_min(lo[d]); hi[d] = wave_max(hi[d]); }
    if (lane_id() == 0) {
#pragma unroll
        for (int d = 0; d < 4; d++) { smn[wave_id()][d] = lo[d]; smx[wave_id()][d] = hi[d]; }
    }
    __syncthreads();
    if (i == 0) {
        float l4[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, h4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int w = 0; w < VOX_TINY / 64; w++)
#pragma unroll
            for (int d = 0; d < 4; d++) { l4[d] = fminf(l4[d], smn[w][d]); h4[d] = fmaxf(h4[d], smx[w][d]); }
        sg.n_valid = n;
        if (n > 0) {
#pragma unroll
            for (int d = 0; d < 3; d++) { sg.bb[d] = f2ord(l4[d]); sg.bb[3 + d] = f2ord(h4[d]); }
        } else {
            sg.bb[0] = sg.bb[1] = sg.bb[2] = 0xFFFFFFFFu; sg.bb[3] = sg.bb[4] = sg.bb[5] = 0u;
            l4[3] = h4[3] = 0.f;
        }
        vox_setup_math(sg, a.leaf, a.seg_cap, l4[3], h4[3], a.bin_pts, a.bin_max);
    }
    __syncthreads();
    if (sg.overflow || n == 0) {                     // PCL's overflow rule: output = input, in input order
        if (in_range) a.out[i] = p;
        if (i == 0) { a.hdr[0] = n; a.hdr[1] = n > 0 ? 1 : 0; }
        return;
    }
    const unsigned key = in_range ? vox_key_of_pt(sg, p) : 0xFFFFFFFFu;
    const unsigned long long mine = in_range ? (((unsigned long long)key << 32) | (unsigned)i) : ~0ull;
    sk[i] = mine;
    if (in_range) a.keys[i] = (int)key;
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < n; j++) rank += sk[j] < mine ? 1 : 0;          // every lane reads the same word: a broadcast
    if (in_range) ssorted[rank] = mine;                                // keys are unique (input index in the low word)
    __syncthreads();
    const unsigned k = in_range ? (unsigned)(ssorted[i] >> 32) : 0u;
    const bool head = in_range && (i == 0 || (unsigned)(ssorted[i - 1] >> 32) != k);
    int nvox;
    const int v = block_excl_scan<VOX_TINY>(head ? 1 : 0, ws, &nvox);
    if (head) {
        const VoxFx f = vox_fx_of(sg);
        unsigned long long sx = 0ull, sy = 0ull, sz = 0ull, si = 0ull;
        unsigned cnt = 0u;
        for (int q = i; q < n && (unsigned)(ssorted[q] >> 32) == k; q++) {
            unsigned long long t[4];
            (void)vox_fx_point(f, spt[(unsigned)ssorted[q]], t);
            sx += t[0]; sy += t[1]; sz += t[2]; si += t[3]; cnt++;
        }
        a.out[v] = fx_centroid(sg, k, sx, sy, sz, si, cnt);
        a.cells[v] = (int)k; a.counts[v] = (int)cnt;
    }
    if (i == 0) { a.hdr[0] = nvox; a.hdr[1] = 0; }
}

}  // namespace

static VoxArgs make_args(const VoxelPlan& p)
{
    return VoxArgs{p.d_static, p.d_dyn, p.d_grid, p.d_n, p.d_nbits, p.sort.keysA, p.sort.valsA, p.sort.keysB, p.sort.valsB,
                   p.d_blockHeads, p.d_starts, p.d_nout, p.nseg, p.seg_cap, p.nblk_h, p.concat_out ? 1 : 0, p.d_mmPartial, p.nblk_mm,
                   p.d_binCount, p.d_binStart, p.d_cursor, p.d_binVox, p.d_binOut, p.d_bucketed, p.d_staging, p.d_stagingKC, p.h_ncells,
                   p.d_chunkStart, p.d_multiStart, p.d_chunkBin, p.max_chunks, p.d_lightBin, p.d_multiOwner, p.d_chunkTabV, p.d_chunkTabC, p.max_multi,
                   {p.n_host[0], p.n_host[1], p.n_host[2], p.n_host[3]}, (p.use_n_host && p.nseg <= 4) ? 1 : 0,
                   {p.n_dev[0], p.n_dev[1], p.n_dev[2], p.n_dev[3]}, p.bin_pts, p.bin_max, ((p.plan_per_run && p.d_wprefix) || (p.bbox_cached && p.hist_cached)) ? p.d_binCountCached : nullptr, p.d_binCountCached, p.d_planMiss, (p.plan_per_run && p.d_wprefix) ? 1 : 0, p.d_wprefix, VB_CH};
}

void VoxelPlan::set_static(const Ctx& ctx, const VoxSegStatic* host_segs)
{
    if (!h_ncells) {
        LVI_HIP(hipHostMalloc((void**)&h_ncells, sizeof(unsigned long long) * nseg, hipHostMallocDefault));
        for (int s = 0; s < nseg; s++) h_ncells[s] = ~0ull;           // unknown: AUTO starts with the sorted path
    }
    LVI_HIP(hipMemcpyAsync(d_static, host_segs, sizeof(VoxSegStatic) * nseg, hipMemcpyHostToDevice, ctx.stream));
    LVI_HIP(hipStreamSynchronize(ctx.stream));
}

void VoxelPlan::release()
{
    if (h_ncells) (void)hipHostFree(h_ncells);
    h_ncells = nullptr;
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
    cells.resize(g.nvox); counts.resize(g.nvox);
    if (p.last_mode == VOX_BINNED) {
        std::vector<int> bs(g.nbins + 1), bv(g.nbins);
        std::vector<uint2> kc(n_in);
        LVI_HIP(hipMemcpyAsync(bs.data(), p.d_binStart, sizeof(int) * (g.nbins + 1), hipMemcpyDeviceToHost, ctx.stream));
        LVI_HIP(hipMemcpyAsync(bv.data(), p.d_binVox, sizeof(int) * g.nbins, hipMemcpyDeviceToHost, ctx.stream));
        LVI_HIP(hipMemcpyAsync(kc.data(), p.d_stagingKC, sizeof(uint2) * n_in, hipMemcpyDeviceToHost, ctx.stream));
        LVI_HIP(hipStreamSynchronize(ctx.stream));
        int v = 0;
        for (int b = 0; b < g.nbins; b++)
            for (int j = 0; j < bv[b] && v < g.nvox; j++, v++) { cells[v] = (int32_t)kc[bs[b] + j].x; counts[v] = (int32_t)kc[bs[b] + j].y; }
    } else {
        const bool inB = (((g.nbits + 7) >> 3) & 1) != 0;
        std::vector<unsigned> sk(n_in);
        std::vector<int> st(g.nvox + 1);
        LVI_HIP(hipMemcpyAsync(sk.data(), inB ? p.sort.keysB : p.sort.keysA, sizeof(unsigned) * n_in, hipMemcpyDeviceToHost, ctx.stream));
        LVI_HIP(hipMemcpyAsync(st.data(), p.d_starts, sizeof(int) * (g.nvox + 1), hipMemcpyDeviceToHost, ctx.stream));
        LVI_HIP(hipStreamSynchronize(ctx.stream));
        for (int v = 0; v < g.nvox; v++) { cells[v] = (int32_t)sk[st[v]]; counts[v] = st[v + 1] - st[v]; }
    }
    // per-point keys: recompute into keysA (scratch is free once the outputs are written)
    Batch<VoxArgs> B;
    B.a[0] = make_args(p);
    hipLaunchKernelGGL(vox_keys_kernel, dim3(div_up(n_in, 256), 1), dim3(256), 0, ctx.stream, B);
    LVI_HIP(hipGetLastError());
    keys.resize(n_in);
    LVI_HIP(hipMemcpyAsync(keys.data(), p.sort.keysA, sizeof(unsigned) * n_in, hipMemcpyDeviceToHost, ctx.stream));
    LVI_HIP(hipStreamSynchronize(ctx.stream));
}

// =====================================================================================================
// Incremental local map (IncMap, lvi_voxel.hpp)
// =====================================================================================================
namespace {

constexpr unsigned long long INC_EMPTY = ~0ull;
constexpr int INC_BIAS = 1 << 20;                     // voxel coordinates in [-2^20, 2^20): +-209 km at a 0.2 m leaf

struct IncArgs {
    unsigned long long* key[2]; unsigned long long* sums[2]; int* cnt[2]; int* occ[2]; int* nocc;
    unsigned* kfBox; const int* active; int n_active;
    const IncPiece* pieces; const lvi_pt* pool;
    unsigned *keysA, *valsA, *keysB, *valsB; int sort_cap;
    int *d_n, *d_nbits, *status;
    VoxGrid* grid; int* nout; lvi_pt* out[2];
    float leaf[2]; float inv[2]; int fx_k[2]; int H, seg_cap;
};

__device__ __forceinline__ unsigned inc_hash(unsigned long long k, int H)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k & (unsigned)(H - 1);
}

__global__ __launch_bounds__(256) void inc_clear_kernel(IncArgs a)
{
    const int w = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.H; i += gridDim.x * 256) {
        a.key[w][i] = INC_EMPTY; a.cnt[w][i] = 0;
        a.sums[w][4 * (size_t)i] = 0ull; a.sums[w][4 * (size_t)i + 1] = 0ull; a.sums[w][4 * (size_t)i + 2] = 0ull; a.sums[w][4 * (size_t)i + 3] = 0ull;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.nocc[w] = 0; if (w == 0) a.status[0] = 0; }
}

// blockIdx.y = piece; every point of the piece goes through transformPointCloud's expression (as kf_assemble) and into its voxel
__global__ __launch_bounds__(256) void inc_apply_kernel(IncArgs a)
{
    const IncPiece pc = a.pieces[blockIdx.y];
    const int w = pc.which;
    const float inv = a.inv[w];
    const double leaf = (double)a.leaf[w];
    const int k = a.fx_k[w];
    const lvi_pt* __restrict__ in = a.pool + pc.in_off;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, imax = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < pc.n; i += gridDim.x * 256) {
        const lvi_pt p = in[i];
        lvi_pt q;
        q.x = pc.A[0] * p.x + pc.A[1] * p.y + pc.A[2] * p.z + pc.A[3];            // pointAssociateToMap / transformPointCloud :339-366
        q.y = pc.A[4] * p.x + pc.A[5] * p.y + pc.A[6] * p.z + pc.A[7];
        q.z = pc.A[8] * p.x + pc.A[9] * p.y + pc.A[10] * p.z + pc.A[11];
        q.intensity = p.intensity;
        const int c0 = (int)floorf(mul_rn(q.x, inv)), c1 = (int)floorf(mul_rn(q.y, inv)), c2 = (int)floorf(mul_rn(q.z, inv));
        if (pc.sign > 0) {
            mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z);
            mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z);
            imax = fmaxf(imax, fabsf(q.intensity));
        }
        if (c0 < -INC_BIAS || c0 >= INC_BIAS || c1 < -INC_BIAS || c1 >= INC_BIAS || c2 < -INC_BIAS || c2 >= INC_BIAS || !(fabsf(q.intensity) < 256.f)) {
            atomicOr(a.status, !(fabsf(q.intensity) < 256.f) ? INC_ERR_INTENSITY : INC_ERR_RANGE);
            continue;
        }
        const unsigned long long key = ((unsigned long long)(c2 + INC_BIAS) << 42) | ((unsigned long long)(c1 + INC_BIAS) << 21) | (unsigned long long)(c0 + INC_BIAS);
        unsigned h = inc_hash(key, a.H);
        int slot = -1;
        for (int probe = 0; probe < a.H; probe++) {
            const unsigned long long cur = a.key[w][h];
            if (cur == key) { slot = (int)h; break; }
            if (cur == INC_EMPTY) {
                const unsigned long long old = atomicCAS(&a.key[w][h], INC_EMPTY, key);
                if (old == INC_EMPTY) { slot = (int)h; a.occ[w][atomicAdd(&a.nocc[w], 1)] = slot; break; }
                if (old == key) { slot = (int)h; break; }
            }
            h = (h + 1) & (unsigned)(a.H - 1);
        }
        if (slot < 0) { atomicOr(a.status, INC_ERR_FULL); continue; }
        unsigned long long v[4] = {fx_xyz(q.x, c0, leaf, k), fx_xyz(q.y, c1, leaf, k), fx_xyz(q.z, c2, leaf, k), fx_int(q.intensity, 37 - 8)};
        unsigned long long* sm = a.sums[w] + 4 * (size_t)slot;
#pragma unroll
        for (int j = 0; j < 4; j++) atomicAdd(&sm[j], pc.sign > 0 ? v[j] : (0ull - v[j]));
        atomicAdd(&a.cnt[w][slot], pc.sign);
    }
    if (pc.sign > 0) {
        // map-frame bbox of this keyframe cloud (folded over the active keys at every emission)
        __shared__ float smn[4][3], smx[4][3], sim[4];
#pragma unroll
        for (int d = 0; d < 3; d++) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
        imax = wave_max(imax);
        if (lane_id() == 0) { for (int d = 0; d < 3; d++) { smn[wave_id()][d] = mn[d]; smx[wave_id()][d] = mx[d]; } sim[wave_id()] = imax; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int d = threadIdx.x;
            const float lo = fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d])), hi = fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d]));
            unsigned* box = a.kfBox + ((size_t)pc.kf * 2 + w) * 8;
            if (lo <= hi) { atomicMin(&box[d], f2ord(lo)); atomicMax(&box[3 + d], f2ord(hi)); }
        }
    }
}

// per keyframe record reset before it is (re)added: min = +inf, max = -inf, n = points
__global__ void inc_box_reset_kernel(IncArgs a, int n_pieces)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const IncPiece pc = a.pieces[i];
    if (pc.sign <= 0) return;
    unsigned* box = a.kfBox + ((size_t)pc.kf * 2 + pc.which) * 8;
    box[0] = box[1] = box[2] = 0xFFFFFFFFu; box[3] = box[4] = box[5] = 0u; box[6] = (unsigned)pc.n; box[7] = 0u;
}

// one wavefront per map kind: bbox of the active keyframes -> VoxGrid (the geometry vox_setup derives from the fused cloud)
__global__ __launch_bounds__(64) void inc_setup_kernel(IncArgs a, int bin_pts, int bin_max)
{
    const int w = blockIdx.x;
    VoxGrid& g = a.grid[w];
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    long long n = 0;
    for (int i = threadIdx.x; i < a.n_active; i += 64) {
        const unsigned* box = a.kfBox + ((size_t)a.active[i] * 2 + w) * 8;
        if (box[6] == 0u) continue;
#pragma unroll
        for (int d = 0; d < 3; d++) { lo[d] = min(lo[d], box[d]); hi[d] = max(hi[d], box[3 + d]); }
        n += box[6];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) { lo[d] = min(lo[d], (unsigned)__shfl_xor((int)lo[d], o, 64)); hi[d] = max(hi[d], (unsigned)__shfl_xor((int)hi[d], o, 64)); }
        n += __shfl_xor(n, o, 64);
    }
    if (threadIdx.x != 0) return;
    g.n_valid = (int)min(n, (long long)0x7fffffff);
    if (n > 0) { for (int d = 0; d < 3; d++) { g.bb[d] = lo[d]; g.bb[3 + d] = hi[d]; } }
    else { g.bb[0] = g.bb[1] = g.bb[2] = 0xFFFFFFFFu; g.bb[3] = g.bb[4] = g.bb[5] = 0u; }
    vox_setup_math(g, a.leaf[w], a.seg_cap, 0.f, 255.f, bin_pts, bin_max);         // intensities were checked to be < 256: fx_ki = 29
    if (g.overflow) atomicOr(a.status, INC_ERR_OVERFLOW);
    a.d_nbits[w] = g.nbits;
    a.d_n[w] = 0;
}

// (idx under the current grid, slot) of every live voxel, in any order (the sort fixes it; idx is unique per voxel)
__global__ __launch_bounds__(256) void inc_keys_kernel(IncArgs a)
{
    const int w = blockIdx.y;
    const VoxGrid& g = a.grid[w];
    const int nocc = a.nocc[w];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < nocc; j += gridDim.x * 256) {
        const int slot = a.occ[w][j];
        if (a.cnt[w][slot] <= 0) continue;
        const unsigned long long key = a.key[w][slot];
        const int c0 = (int)(key & 0x1FFFFFull) - INC_BIAS, c1 = (int)((key >> 21) & 0x1FFFFFull) - INC_BIAS, c2 = (int)(key >> 42) - INC_BIAS;
        const unsigned idx = (unsigned)(c0 - g.min_b[0]) + (unsigned)(c1 - g.min_b[1]) * g.mul1 + (unsigned)(c2 - g.min_b[2]) * g.mul2;
        const int pos = atomicAdd(&a.d_n[w], 1);
        if (pos >= a.sort_cap) { atomicOr(a.status, INC_ERR_FULL); continue; }
        a.keysA[(size_t)w * a.sort_cap + pos] = idx; a.valsA[(size_t)w * a.sort_cap + pos] = (unsigned)slot;
    }
}

__global__ __launch_bounds__(256) void inc_out_kernel(IncArgs a)
{
    const int w = blockIdx.y;
    VoxGrid& g = a.grid[w];
    const int n = min(a.d_n[w], a.sort_cap);
    const bool inB = rs_result_in_B(a.d_nbits[w]);
    const unsigned* keys = (inB ? a.keysB : a.keysA) + (size_t)w * a.sort_cap;
    const unsigned* vals = (inB ? a.valsB : a.valsA) + (size_t)w * a.sort_cap;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
        const int slot = (int)vals[r];
        const unsigned long long* sm = a.sums[w] + 4 * (size_t)slot;
        a.out[w][r] = fx_centroid(g, keys[r], sm[0], sm[1], sm[2], sm[3], (unsigned)a.cnt[w][slot]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g.nvox = n; g.out_off = 0; a.nout[w] = n;
        if (w == 1) a.nout[2] = a.d_n[0] + n;
    }
}

IncArgs inc_args(const IncMap& m, const float leaf[2])
{
    IncArgs a{};
    for (int w = 0; w < 2; w++) {
        a.key[w] = m.key[w]; a.sums[w] = m.sums[w]; a.cnt[w] = m.cnt[w]; a.occ[w] = m.occ[w];
        a.leaf[w] = leaf[w]; a.inv[w] = 1.0f / leaf[w];
        int ex = 0; (void)frexpf(2.0f * leaf[w], &ex); a.fx_k[w] = 37 - ex;          // vox_fx_setup
    }
    a.nocc = m.d_nocc; a.kfBox = m.kfBox; a.active = m.d_active; a.pieces = m.d_pieces;
    a.keysA = m.sort.keysA; a.valsA = m.sort.valsA; a.keysB = m.sort.keysB; a.valsB = m.sort.valsB; a.sort_cap = m.sort.seg_cap;
    a.d_n = m.d_n; a.d_nbits = m.d_nbits; a.status = m.d_status; a.H = m.H;
    return a;
}

}  // namespace

void incmap_clear(const Ctx& ctx, const IncMap& m)
{
    const float leaf[2] = {1.f, 1.f};
    IncArgs a = inc_args(m, leaf);
    LVI_LAUNCH(ctx, "inc_clear", 48.0 * m.H * 2, hipLaunchKernelGGL(inc_clear_kernel, dim3(std::min(div_up(m.H, 256), 2048), 2), dim3(256), 0, ctx.stream, a));
}

void incmap_apply(const Ctx& ctx, const IncMap& m, const lvi_pt* pool, int n_pieces, int max_n, const float leaf[2])
{
    if (n_pieces <= 0) return;
    IncArgs a = inc_args(m, leaf);
    a.pool = pool;
    LVI_LAUNCH(ctx, "inc_box_reset", 0, hipLaunchKernelGGL(inc_box_reset_kernel, dim3(div_up(n_pieces, 64)), dim3(64), 0, ctx.stream, a, n_pieces));
    LVI_LAUNCH(ctx, "inc_apply", 0, hipLaunchKernelGGL(inc_apply_kernel, dim3(std::max(1, std::min(div_up(max_n, 256), 32)), n_pieces), dim3(256), 0, ctx.stream, a));
}

void incmap_emit(const Ctx& ctx, const IncMap& m, int n_active, const float leaf[2], VoxGrid* grid, int* nout, lvi_pt* outC, lvi_pt* outS, int seg_cap)
{
    IncArgs a = inc_args(m, leaf);
    a.n_active = n_active; a.grid = grid; a.nout = nout; a.out[0] = outC; a.out[1] = outS; a.seg_cap = seg_cap;
    const VoxelPlan defaults;
    LVI_LAUNCH(ctx, "inc_setup", 0, hipLaunchKernelGGL(inc_setup_kernel, dim3(2), dim3(64), 0, ctx.stream, a, defaults.bin_pts, defaults.bin_max));
    const dim3 gk(std::min(div_up(m.H, 256), 1024), 2);
    LVI_LAUNCH(ctx, "inc_keys", 0, hipLaunchKernelGGL(inc_keys_kernel, gk, dim3(256), 0, ctx.stream, a));
    radix_sort_pairs(ctx, m.sort, m.d_n, m.d_nbits, 4, "inc", 0.0);
    LVI_LAUNCH(ctx, "inc_out", 0, hipLaunchKernelGGL(inc_out_kernel, gk, dim3(256), 0, ctx.stream, a));
}

// What depends on the plan's INPUT alone and is produced where the input is written (upload / assembly) instead of once per
// re-voxelisation: the bbox partial records and — the grid geometry following from the bbox — the points per bin.
void voxel_tiny(const Ctx& ctx, const lvi_pt* in_pinned, int n, float leaf, int seg_cap, int bin_pts, int bin_max, lvi_pt* out_pinned, int* hdr_pinned,
                int* cells_pinned, int* counts_pinned, int* keys_pinned)
{
    VoxTinyArgs a{in_pinned, n, leaf, seg_cap, bin_pts, bin_max, out_pinned, hdr_pinned, cells_pinned, counts_pinned, keys_pinned};
    LVI_LAUNCH(ctx, "vox_tiny", 32.0 * n, hipLaunchKernelGGL(vox_tiny_kernel, dim3(1), dim3(VOX_TINY), 0, ctx.stream, a));
}

void voxel_bbox_pass(const Ctx& ctx, const VoxelPlan& p, const char* tag, double n_hint)
{
    Batch<VoxArgs> B;
    B.a[0] = make_args(p);
    B.a[0].binCountCached = nullptr;
    for (int z = 1; z < MAX_BATCH; z++) B.a[z] = B.a[0];
    char nm[3][48];
    snprintf(nm[0], sizeof(nm[0]), "vox_minmax/%s", tag); snprintf(nm[1], sizeof(nm[1]), "vox_setup/%s", tag); snprintf(nm[2], sizeof(nm[2]), "vb_hist/%s", tag);
    LVI_LAUNCH(ctx, nm[0], 16.0 * n_hint, hipLaunchKernelGGL(vox_minmax_kernel, dim3(p.nblk_mm, p.nseg, 1), dim3(256), 0, ctx.stream, B));
    p.hist_cached = false;
    if (voxel_resolve_mode(p) == VOX_BINNED && p.d_binCountCached) {
        LVI_LAUNCH(ctx, nm[1], 0, hipLaunchKernelGGL(vox_setup_kernel, dim3(p.nseg, 1, 1), dim3(64), 0, ctx.stream, B));
        LVI_LAUNCH(ctx, nm[2], 16.0 * n_hint, hipLaunchKernelGGL(vb_hist_w_kernel, dim3(VB_WG, p.nseg, 1), dim3(256), 0, ctx.stream, B));
        hipLaunchKernelGGL(vb_colscan_kernel, dim3(VB_NB / 16, p.nseg, 1), dim3(256), 0, ctx.stream, B);
        LVI_HIP(hipGetLastError());
        p.hist_cached = true;
    }
}

int voxel_resolve_mode(const VoxelPlan& p)
{
    if (p.mode != VOX_AUTO) return p.mode;
    // hint of the previous batch of this plan (pinned host memory written by vox_setup; a racing read returns the
    // older or the newer value, and either path is correct for any grid)
    int mode = VOX_BINNED;
    for (int s = 0; s < p.nseg; s++)
        // up to four 1024-voxel sub-ranges per bin the binned path still wins (4 M sparse points, 12.5 M cells: 341 vs
        // 416 us); at 100 M cells it is 4x slower than the sort (tools/ubench/sparse_voxel.py)
        if (!p.h_ncells || p.h_ncells[s] > ((unsigned long long)VB_NB << (VB_CL_LOG + 2))) mode = VOX_SORTED;
    return mode;
}

void voxel_downsample_batch(const Ctx& ctx, const VoxelPlan& p, const char* tag, double n_hint)
{
    const VoxelPlan* one = &p;
    voxel_downsample_batch(ctx, &one, 1, tag, n_hint);
}

// S plans of identical shape (the same plan of S batch slots), one launch sequence: blockIdx.z = slot
void voxel_downsample_batch(const Ctx& ctx, const VoxelPlan* const* plans, int S, const char* tag, double n_hint)
{
    const VoxelPlan& p = *plans[0];
    // the realisation must be the same for the whole launch: sorted as soon as one slot asks for it
    int mode = VOX_BINNED;
    for (int z = 0; z < S; z++) if (voxel_resolve_mode(*plans[z]) == VOX_SORTED) mode = VOX_SORTED;
    if (mode == VOX_SORTED && S > 1) {
        // the radix sort is not batched over slots (sparse outdoor grids only): slot by slot
        for (int z = 0; z < S; z++) {
            const int keep = plans[z]->mode;
            const_cast<VoxelPlan*>(plans[z])->mode = VOX_SORTED;
            voxel_downsample_batch(ctx, plans + z, 1, tag, n_hint / S);
            const_cast<VoxelPlan*>(plans[z])->mode = keep;
        }
        return;
    }
    Batch<VoxArgs> B;
    static const int ch_env = getenv("LVI_VB_CH") ? atoi(getenv("LVI_VB_CH")) : 0;
    const int ch = ch_env >= VB_CH && ch_env <= 32768 ? ch_env : VB_CH;
    for (int z = 0; z < S; z++) { B.a[z] = make_args(*plans[z]); B.a[z].ch = ch; plans[z]->last_mode = mode; }
    for (int z = S; z < MAX_BATCH; z++) B.a[z] = B.a[0];
    const VoxArgs& a = B.a[0];
    char nm[16][48];
    const char* base[16] = {"vox_minmax", "vox_setup", "vox_keys", "vox_heads_count", "vox_heads_scan", "vox_heads_assign", "vox_centroid",
                            "vb_hist", "vb_scan", "vb_scatter", "vb_accum", "vb_outscan", "vb_copy", "vb_merge", "vb_light", ""};
    for (int i = 0; i < 15; i++) snprintf(nm[i], sizeof(nm[i]), "%s/%s", base[i], tag);
    // the bbox pass is skipped when every plan of the batch holds the partial records of its (unchanged) input: the raw
    // local map gets them where its points are touched anyway — upload / assembly — instead of once per re-voxelisation
    bool cached = true, per_run = mode == VOX_BINNED;
    for (int z = 0; z < S; z++) { cached = cached && plans[z]->bbox_cached; per_run = per_run && B.a[z].plan_spec != 0; }
    if (per_run) {
        // reference-faithful raw map: bbox + per-bin counts inside this run, one pass (vb_plan), validated by vox_setup
        LVI_LAUNCH(ctx, "vb_plan/map", 16.0 * n_hint, hipLaunchKernelGGL(vb_plan_kernel, dim3(VB_WG, p.nseg, S), dim3(256), 0, ctx.stream, B));
        LVI_LAUNCH(ctx, nm[1], 0, hipLaunchKernelGGL(vox_setup_kernel, dim3(p.nseg, 1, S), dim3(64), 0, ctx.stream, B));
        LVI_LAUNCH(ctx, "vb_hist_w/map", 0, hipLaunchKernelGGL(vb_hist_w_kernel, dim3(VB_WG, p.nseg, S), dim3(256), 0, ctx.stream, B));
        LVI_LAUNCH(ctx, "vb_colscan/map", 0, hipLaunchKernelGGL(vb_colscan_kernel, dim3(VB_NB / 16, p.nseg, S), dim3(256), 0, ctx.stream, B));
    } else {
        for (int z = 0; z < S; z++) B.a[z].plan_spec = 0;
        if (!cached) LVI_LAUNCH(ctx, nm[0], 16.0 * n_hint, hipLaunchKernelGGL(vox_minmax_kernel, dim3(p.nblk_mm, p.nseg, S), dim3(256), 0, ctx.stream, B));
        LVI_LAUNCH(ctx, nm[1], 0, hipLaunchKernelGGL(vox_setup_kernel, dim3(p.nseg, 1, S), dim3(64), 0, ctx.stream, B));
    }
    if (mode == VOX_BINNED) {
        const dim3 gt(div_up(p.seg_cap, VB_STILE), p.nseg, S);
        const dim3 gb(std::max(512, std::min(div_up(p.seg_cap, 2048), VB_ACC_BLOCKS)), p.nseg, S);      // grid-stride over the bins (a workgroup per bin or two: a bin is three dependent steps)
        const dim3 gh2(std::min(div_up(p.seg_cap, VB_TILE), 512), p.nseg, S);
        bool hist_cached = true;
        for (int z = 0; z < S; z++) hist_cached = hist_cached && B.a[z].binCountCached != nullptr;
        if (!hist_cached) {
            for (int z = 0; z < S; z++) B.a[z].binCountCached = nullptr;                  // all slots the same way
            LVI_LAUNCH(ctx, nm[7], 16.0 * n_hint, hipLaunchKernelGGL(vb_hist_kernel, gh2, dim3(256), 0, ctx.stream, B));
        }
        LVI_LAUNCH(ctx, nm[8], 0, hipLaunchKernelGGL(vb_scan_kernel, dim3(p.nseg, 1, S), dim3(256), 0, ctx.stream, B));
        if (hist_cached) LVI_LAUNCH(ctx, nm[9], 32.0 * n_hint, hipLaunchKernelGGL(vb_scatter_det_kernel, dim3(VB_WG, p.nseg, S), dim3(256), 0, ctx.stream, B));
        else LVI_LAUNCH(ctx, nm[9], 32.0 * n_hint, hipLaunchKernelGGL(vb_scatter_kernel, gt, dim3(256), 0, ctx.stream, B));
        // grid-stride over the chunks: at most ceil(n / VB_CH) + bins of them exist; a small plan (ring / scan grids: 64 bins) gets
        // a small grid — every workgroup of this kernel owns 40 KB of LDS, and thousands of idle ones cost 40 us of dispatch
        const dim3 ga(std::max(64, std::min(2 * div_up(p.seg_cap, VB_CH) + 64, 2048)), p.nseg, S);
        // 256 threads: 128 / 512 / 1024 measured 172 / 163 / 231 us instead of 136 for four slots of the 4.87 M-point map
        LVI_LAUNCH(ctx, nm[10], 16.0 * n_hint, hipLaunchKernelGGL(vb_accum_kernel<256>, ga, dim3(256), 0, ctx.stream, B));
        LVI_LAUNCH(ctx, nm[13], 0, hipLaunchKernelGGL(vb_merge_kernel, dim3(std::max(1, std::min(p.max_multi, VB_ACC_BLOCKS)), p.nseg, S), dim3(256), 0, ctx.stream, B));
        LVI_LAUNCH(ctx, nm[11], 0, hipLaunchKernelGGL(vb_outscan_kernel, dim3(1, 1, S), dim3(256), 0, ctx.stream, B));
        LVI_LAUNCH(ctx, nm[12], 0, hipLaunchKernelGGL(vb_copy_kernel, gb, dim3(256), 0, ctx.stream, B));
        return;
    }
    (void)a;
    const dim3 gp(div_up(p.seg_cap, 256), p.nseg), gh(p.nblk_h, p.nseg);
    LVI_LAUNCH(ctx, nm[2], 24.0 * n_hint, hipLaunchKernelGGL(vox_keys_kernel, gp, dim3(256), 0, ctx.stream, B));
    radix_sort_pairs(ctx, p.sort, p.d_n, p.d_nbits, 4, tag, n_hint);
    LVI_LAUNCH(ctx, nm[3], 4.0 * n_hint, hipLaunchKernelGGL(vox_heads_count_kernel, gh, dim3(256), 0, ctx.stream, B));
    LVI_LAUNCH(ctx, nm[4], 0, hipLaunchKernelGGL(vox_heads_scan_kernel, dim3(1), dim3(256), 0, ctx.stream, B));
    LVI_LAUNCH(ctx, nm[5], 4.0 * n_hint, hipLaunchKernelGGL(vox_heads_assign_kernel, gh, dim3(256), 0, ctx.stream, B));
    const dim3 gc(std::max(1, std::min(div_up(p.seg_cap, 256 / 8), 8192)), p.nseg);
    if (p.centroid_lanes >= 32)
        LVI_LAUNCH(ctx, nm[6], 20.0 * n_hint, hipLaunchKernelGGL(vox_centroid_kernel<32>, gc, dim3(256), 0, ctx.stream, B));
    else
        LVI_LAUNCH(ctx, nm[6], 20.0 * n_hint, hipLaunchKernelGGL(vox_centroid_kernel<8>, gc, dim3(256), 0, ctx.stream, B));
}

}  // namespace lvi
