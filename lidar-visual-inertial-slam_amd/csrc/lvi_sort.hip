// Stable LSD radix sort (8-bit digits) of (u32 key, u32 value) pairs for gfx950.
//
// One pass = three launches over all segments at once (blockIdx.y = segment):
//   rs_hist     per-tile digit histogram (LDS atomics)                 reads 4 B/pair
//   rs_scan     per-digit exclusive scan over the tiles (in place)     tiny
//   rs_scatter  wave64 ballot "match" ranking → tile staged in LDS in digit order →
//               coalesced runs to HBM                                  reads 8 B, writes 8 B/pair
// Segment lengths and key widths are read from device memory, so a whole voxel-grid or
// grid-index build is enqueued without a host round trip; passes above a segment's key width
// exit at once.  Stability (equal keys keep input order) is what makes the voxel centroid
// sums deterministic: points of one voxel are summed in input order.
#include "lvi_sort.hpp"

namespace lvi {

namespace {

struct SortArgs {
    unsigned *keysA, *valsA, *keysB, *valsB, *hist, *digitTotal;
    int seg_cap, nblk;
};

template <int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(SortArgs a, const int* __restrict__ d_n, const int* __restrict__ d_nbits, int pass)
{
    constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
    const int s = blockIdx.y;
    const int shift = pass * 8;
    if (shift >= d_nbits[s]) return;
    const int n = d_n[s];
    const int base = blockIdx.x * RS_TILE;
    if (base >= n) return;
    const unsigned* keys = ((pass & 1) ? a.keysB : a.keysA) + (size_t)s * a.seg_cap;
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    unsigned kk[RS_ITEMS];                                  // unconditional loads (clamped index): all of a thread's keys in flight together
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) kk[i] = keys[min(base + i * RS_THREADS + (int)threadIdx.x, n - 1)];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const int idx = base + i * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(kk[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    a.hist[((size_t)s * 256 + threadIdx.x) * a.nblk + blockIdx.x] = h[threadIdx.x];
}

// grid (256 digits, nseg): exclusive scan of hist[s][d][0..tiles) in place, total to digitTotal
template <int RS_ITEMS>
__global__ __launch_bounds__(256) void rs_scan_kernel(SortArgs a, const int* __restrict__ d_n, const int* __restrict__ d_nbits, int pass)
{
    constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
    const int s = blockIdx.y, d = blockIdx.x;
    if (pass * 8 >= d_nbits[s]) return;
    const int n = d_n[s];
    const int nt = (n + RS_TILE - 1) / RS_TILE;
    unsigned* row = a.hist + ((size_t)s * 256 + d) * a.nblk;
    __shared__ int ws[8];
    int carry = 0;
    for (int c = 0; c < nt; c += 256) {
        const int i = c + threadIdx.x;
        const int v = (i < nt) ? (int)row[i] : 0;
        int tot;
        const int ex = block_excl_scan<256>(v, ws, &tot);
        if (i < nt) row[i] = (unsigned)(carry + ex);
        carry += tot;
    }
    if (threadIdx.x == 0) a.digitTotal[s * 256 + d] = (unsigned)carry;
}

template <int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(SortArgs a, const int* __restrict__ d_n, const int* __restrict__ d_nbits, int pass)
{
    constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
    const int s = blockIdx.y;
    const int shift = pass * 8;
    if (shift >= d_nbits[s]) return;
    const int n = d_n[s];
    const int base = blockIdx.x * RS_TILE;
    if (base >= n) return;
    const size_t so = (size_t)s * a.seg_cap;
    const unsigned* __restrict__ srcK = ((pass & 1) ? a.keysB : a.keysA) + so;
    const unsigned* __restrict__ srcV = ((pass & 1) ? a.valsB : a.valsA) + so;
    unsigned* __restrict__ dstK = ((pass & 1) ? a.keysA : a.keysB) + so;
    unsigned* __restrict__ dstV = ((pass & 1) ? a.valsA : a.valsB) + so;

    constexpr int NW = RS_THREADS / 64;
    __shared__ unsigned sk[RS_TILE], sv[RS_TILE];
    __shared__ unsigned waveCnt[NW][256];
    __shared__ unsigned digitBase[256];
    __shared__ int lbase[256];
    __shared__ int ws[8];

    const int tid = threadIdx.x, w = wave_id(), l = lane_id();
    {   // global base of every digit for this tile
        int tot;
        const int gb = block_excl_scan<256>((int)a.digitTotal[s * 256 + tid], ws, &tot);
        digitBase[tid] = (unsigned)gb + a.hist[((size_t)s * 256 + tid) * a.nblk + blockIdx.x];
#pragma unroll
        for (int q = 0; q < NW; q++) waveCnt[q][tid] = 0;
    }
    __syncthreads();

    unsigned k[RS_ITEMS], v[RS_ITEMS];
    unsigned short r[RS_ITEMS];
    const int cbase = base + w * (RS_ITEMS * 64);
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {                    // unconditional loads (clamped index), all in flight before the ranking below
        const int ic = min(cbase + i * 64 + l, n - 1);
        k[i] = srcK[ic]; v[i] = srcV[ic];
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const int idx = cbase + i * 64 + l;
        const bool valid = idx < n;
        if (!valid) { k[i] = 0u; v[i] = 0u; }
        const unsigned d = (k[i] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        if (valid) {
            const unsigned prior = waveCnt[w][d];
            r[i] = (unsigned short)(prior + __popcll(peers & lt));
            if ((peers & lt) == 0) waveCnt[w][d] = prior + __popcll(peers);      // lowest peer lane
        } else {
            r[i] = 0;
        }
    }
    __syncthreads();
    {   // per digit: offsets of the waves inside the tile, then tile-local digit bases
        unsigned off = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) { const unsigned c = waveCnt[q][tid]; waveCnt[q][tid] = off; off += c; }
        int tot;
        lbase[tid] = block_excl_scan<256>((int)off, ws, &tot);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const int idx = cbase + i * 64 + l;
        if (idx < n) {
            const unsigned d = (k[i] >> shift) & 255u;
            const int pos = lbase[d] + (int)waveCnt[w][d] + (int)r[i];
            sk[pos] = k[i]; sv[pos] = v[i];
        }
    }
    __syncthreads();
    const int cnt = min(RS_TILE, n - base);
    for (int j = tid; j < cnt; j += RS_THREADS) {
        const unsigned key = sk[j];
        const unsigned d = (key >> shift) & 255u;
        const unsigned g = digitBase[d] + (unsigned)(j - lbase[d]);
        dstK[g] = key; dstV[g] = sv[j];
    }
}

}  // namespace

void radix_sort_pairs(const Ctx& ctx, const SortPlan& p, const int* d_n, const int* d_nbits, int max_passes, const char* tag, double n_hint)
{
    SortArgs a{p.keysA, p.valsA, p.keysB, p.valsB, p.hist, p.digitTotal, p.seg_cap, p.nblk};
    const dim3 gt(p.nblk, p.nseg), gs(256, p.nseg);
    char nm[3][48];
    snprintf(nm[0], sizeof(nm[0]), "rs_hist/%s", tag);
    snprintf(nm[1], sizeof(nm[1]), "rs_scan/%s", tag);
    snprintf(nm[2], sizeof(nm[2]), "rs_scatter/%s", tag);
    for (int pass = 0; pass < max_passes; pass++) {
        if (p.items == RS_ITEMS_SMALL) {
            LVI_LAUNCH(ctx, nm[0], 4.0 * n_hint, hipLaunchKernelGGL(rs_hist_kernel<RS_ITEMS_SMALL>, gt, dim3(RS_THREADS), 0, ctx.stream, a, d_n, d_nbits, pass));
            LVI_LAUNCH(ctx, nm[1], 0.0, hipLaunchKernelGGL(rs_scan_kernel<RS_ITEMS_SMALL>, gs, dim3(256), 0, ctx.stream, a, d_n, d_nbits, pass));
            LVI_LAUNCH(ctx, nm[2], 16.0 * n_hint, hipLaunchKernelGGL(rs_scatter_kernel<RS_ITEMS_SMALL>, gt, dim3(RS_THREADS), 0, ctx.stream, a, d_n, d_nbits, pass));
        } else {
            LVI_LAUNCH(ctx, nm[0], 4.0 * n_hint, hipLaunchKernelGGL(rs_hist_kernel<RS_ITEMS>, gt, dim3(RS_THREADS), 0, ctx.stream, a, d_n, d_nbits, pass));
            LVI_LAUNCH(ctx, nm[1], 0.0, hipLaunchKernelGGL(rs_scan_kernel<RS_ITEMS>, gs, dim3(256), 0, ctx.stream, a, d_n, d_nbits, pass));
            LVI_LAUNCH(ctx, nm[2], 16.0 * n_hint, hipLaunchKernelGGL(rs_scatter_kernel<RS_ITEMS>, gt, dim3(RS_THREADS), 0, ctx.stream, a, d_n, d_nbits, pass));
        }
    }
}

}  // namespace lvi
