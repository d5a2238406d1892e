// Map-side kernels of the lidar path for gfx950 (SURVEY §8 a-4(map) … a-10):
//   map build   extractCloud's two VoxelGrid calls + the replacement of the two
//               KdTreeFLANN::setInputCloud calls (mapOptimization.cpp:958-965, 1322-1323)
//   residuals   cornerOptimization + surfOptimization, one thread per feature (:1006-1167)
//   solve       combineOptimizationCoeffs + LMOptimization on the device (:1169-1313)
//   finish      transformUpdate (:1345-1375)
//
// Exact 5-NN without a kd-tree: both callers discard a query unless its 5th neighbour is closer
// than 1 m (sqDis[4] < 1.0, :1025,1121).  The DS map is binned into a uniform grid of 0.5 m cells
// (1 m or more when 0.5 m would need more than max_cells; cell ids computed in double so that
// floor() is exact); every map point with squared distance < 1 to a query lies in the 5x5x5 (3x3x3)
// block of cells around it, and inside the block only in the cells the unit ball reaches, so scanning
// those yields exactly FLANN's answer for every query the reference accepts, and "fewer than 5 within
// 1 m" for the rest.
// Distances are ((dx*dx)+dy*dy)+dz*dz in f32 without FMA, as FLANN's L2_Simple computes them.
//
// The Gauss-Newton loop never returns to the host, one launch per iteration: the 27 sums of AtA/AtB are formed in double
// (cv::gemm accumulates f32 products in double) per 16 features, turned into exact fixed point and added to sharded
// accumulators; the NEXT launch reads the totals and every workgroup runs the 6x6 end of the iteration (QR solve, the
// iteration-0 eigen analysis, pose update, convergence test) on them — no exchange inside a launch.
#include <cstdlib>
#include <type_traits>

#include "lvi_lidar.hpp"

namespace lvi {

namespace {

// ------------------------------------------------------------------------------------------- grid index
struct GridArgs {
    GridIndex::Meta* meta[2];
    const VoxGrid* vox;            // voxMap.d_grid (bbox of the raw map)
    const int* nout;               // voxMap.d_nout
    const lvi_pt* ds[2];
    int* cell_start[2];
    lvi_pt* sorted[2];
    int* count[2];                 // [max_cells + 1] points per cell, zero between builds
    int* blockSum[2];              // [GRID_SCAN_BLOCKS]
    int cap, max_cells;
    int* d_status;
};

// d_status[1] is the status word of the map build: every build rewrites it (the scan-side word d_status[0] is cleared by
// the scan upload, which may come before or after the build on the stream)
__device__ int grid_meta_one(const GridArgs& a, int w, GridIndex::Meta& m)
{
    const int n = a.nout[w];
    m.n = n; m.ok = 0; m.R = 1; m.edge = 0.5; m.inv_edge = 2.0;
    m.dim[0] = m.dim[1] = m.dim[2] = 1; m.ncells = 1; m.origin[0] = m.origin[1] = m.origin[2] = 0.0;
    if (n <= 0 || a.vox[w].n_valid == 0) return 0;
    double lo[3], ext[3];
    for (int d = 0; d < 3; d++) {
        const double mn = floor((double)ord2f(a.vox[w].bb[d])), mx = floor((double)ord2f(a.vox[w].bb[3 + d]));
        lo[d] = mn - 1.0;                       // one cell of padding on every side
        ext[d] = mx - mn + 3.0;
    }
    int c = 1;                                  // cell edge in half metres, integer >= 1
    for (;; c++) {
        const double e = 0.5 * c;
        const double nx = ceil(ext[0] / e), ny = ceil(ext[1] / e), nz = ceil(ext[2] / e);
        if (nx * ny * nz <= (double)a.max_cells) { m.dim[0] = (int)nx; m.dim[1] = (int)ny; m.dim[2] = (int)nz; break; }
        if (c > 1 << 20) return DEV_ERR_GRID_TOO_LARGE;
    }
    // cell = floor((p - lo) / edge); a neighbour within 1 m is at most R cells away on every axis
    m.origin[0] = lo[0]; m.origin[1] = lo[1]; m.origin[2] = lo[2];
    m.ncells = m.dim[0] * m.dim[1] * m.dim[2];
    m.edge = 0.5 * c; m.inv_edge = 1.0 / m.edge; m.R = c == 1 ? 2 : 1;
    m.ok = 1;
    return 0;
}

__device__ __forceinline__ void cell_of(const GridIndex::Meta& m, float x, float y, float z, int c[3])
{
    // the same expression places the map points (grid_keys) and the queries (knn5_search_group)
    c[0] = (int)floor(((double)x - m.origin[0]) * m.inv_edge);
    c[1] = (int)floor(((double)y - m.origin[1]) * m.inv_edge);
    c[2] = (int)floor(((double)z - m.origin[2]) * m.inv_edge);
}

__device__ __forceinline__ int cell_id_of(const GridIndex::Meta& m, const lvi_pt& p)
{
    int c[3];
    cell_of(m, p.x, p.y, p.z, c);
#pragma unroll
    for (int d = 0; d < 3; d++) c[d] = min(max(c[d], 0), m.dim[d] - 1);
    return (c[2] * m.dim[1] + c[1]) * m.dim[0] + c[0];
}

// The index is a counting sort by cell WITHOUT a stable order: the 5-NN result is a minimum over a total order
// of (distance, index) keys, so the order of the points inside a cell cannot change it.  count → exclusive scan
// (two kernels, GRID_SCAN_BLOCKS chunks) → scatter.  Every kernel is a grid-stride loop over device-side counts: the
// launch geometry does not depend on the (device-only) map size.
// Round 3: (1) the grid's geometry is derived by every workgroup of the count kernel itself (it follows from the map's bounding box:
// a few hundred cycles; the one-thread kernel in front of the chain was 5 us of every map build) — the first workgroup of a
// segment leaves it in global memory for the kernels behind; (2) no cursor array: the scan writes start(c) into cell_start[c + 1]
// and the scatter advances THAT word, which ends as start(c) + count(c) = start(c + 1) — the array the search reads, with
// cell_start[0] = 0 (2.6 MB less written per build, 2.6 MB less memory per slot); (3) the geometry is held in registers inside
// the point loops (behind an atomic the compiler re-read origin / inv_edge / dim from global memory for every point).
constexpr int GRID_PT_BLOCKS = 512;
constexpr int GRID_SCAN_BLOCKS = 1024;

__global__ __launch_bounds__(256) void grid_count_kernel(Batch<GridArgs> B_)
{
    const GridArgs& a = B_.a[blockIdx.z];
    const int w = blockIdx.y;
    __shared__ GridIndex::Meta sm[2];
    __shared__ int serr[2];
    const bool lead = blockIdx.x == 0 && w == 0;                  // this workgroup also reports the build's status word
    if (threadIdx.x == 0) serr[w] = grid_meta_one(a, w, sm[w]);
    if (threadIdx.x == 64 && lead) serr[1] = grid_meta_one(a, 1, sm[1]);
    __syncthreads();
    const GridIndex::Meta m = sm[w];
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.meta[w] = m;
    if (lead && threadIdx.x == 0) a.d_status[1] = serr[0] | serr[1];
    if (!m.ok) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m.n; i += gridDim.x * 256) atomicAdd(&a.count[w][cell_id_of(m, a.ds[w][i])], 1);
}

// chunk b of the ncells entries: its total
__global__ __launch_bounds__(256) void grid_scan_sum_kernel(Batch<GridArgs> B_)
{
    const GridArgs& a = B_.a[blockIdx.z];
    const int w = blockIdx.y;
    const int ncells = a.meta[w]->ncells, ok = a.meta[w]->ok;
    const int L = (ncells + GRID_SCAN_BLOCKS - 1) / GRID_SCAN_BLOCKS;
    const int c0 = blockIdx.x * L, c1 = min(ncells, c0 + L);
    int v = 0;
    if (ok) {
        for (int c = c0 + threadIdx.x; c < c1; c += 8 * 256) {      // eight loads in flight
            int t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = a.count[w][min(c + u * 256, max(c1 - 1, 0))];
#pragma unroll
            for (int u = 0; u < 8; u++) v += c + u * 256 < c1 ? t[u] : 0;
        }
    }
    __shared__ int ws[8];
    int tot;
    (void)block_excl_scan<256>(v, ws, &tot);
    if (threadIdx.x == 0) a.blockSum[w][blockIdx.x] = tot;
}

// cell_start[0] = 0, cell_start[c + 1] = points in cells < c (the scatter's cursor of cell c; start(c + 1) once the scatter is through);
// count back to zero
__global__ __launch_bounds__(256) void grid_scan_apply_kernel(Batch<GridArgs> B_)
{
    const GridArgs& a = B_.a[blockIdx.z];
    const int w = blockIdx.y;
    const int ncells = a.meta[w]->ncells, ok = a.meta[w]->ok;
    const int L = (ncells + GRID_SCAN_BLOCKS - 1) / GRID_SCAN_BLOCKS;
    const int c0 = blockIdx.x * L, c1 = min(ncells, c0 + L);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.cell_start[w][0] = 0;
    if (c0 >= c1) return;
    __shared__ int ws[8];
    int carry;
    {
        int v = 0;
        for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) v += a.blockSum[w][j];
        (void)block_excl_scan<256>(v, ws, &carry);
    }
    int* __restrict__ cnt = a.count[w];
    int* __restrict__ cs = a.cell_start[w];
    for (int base = c0; base < c1; base += 256) {
        const int c = base + threadIdx.x;
        const bool live = c < c1 && ok;
        const int v = live ? cnt[c] : 0;
        int tot;
        const int ex = carry + block_excl_scan<256>(v, ws, &tot);
        if (c < c1) cs[c + 1] = ex;
        if (live && v) cnt[c] = 0;                                 // (most cells are empty: nothing to write back)
        carry += tot;
    }
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(Batch<GridArgs> B_)
{
    const GridArgs& a = B_.a[blockIdx.z];
    const int w = blockIdx.y;
    const GridIndex::Meta m = *a.meta[w];
    if (!m.ok) return;
    int* __restrict__ cs = a.cell_start[w];
    lvi_pt* __restrict__ sorted = a.sorted[w];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m.n; i += gridDim.x * 256) {
        lvi_pt p = a.ds[w][i];
        const int pos = atomicAdd(&cs[cell_id_of(m, p) + 1], 1);
        p.intensity = __int_as_float(i);        // original index in laserCloud*FromMapDS
        sorted[pos] = p;
    }
}

// ------------------------------------------------------------------------------------------- 5-NN
struct Knn5 { float d[5]; int i[5]; };

// Private top-5 of a lane, ascending by (distance, index) as ONE 64-bit key per entry: squared distances are
// >= 0, so their IEEE bit patterns order like unsigned integers; the index breaks exact ties.
struct KnnKeys { unsigned long long k[5]; };
constexpr unsigned long long KNN_EMPTY = 0x7F8000007FFFFFFFull;          // (+inf, INT_MAX)
__device__ __forceinline__ unsigned long long knn_key(float dist, int idx) { return ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)idx; }
__device__ __forceinline__ void knn_insert(KnnKeys& r, unsigned long long key)
{
    // branch-free insertion into the sorted list: entry j becomes min(max(old[j-1], key), old[j])
    const unsigned long long o0 = r.k[0], o1 = r.k[1], o2 = r.k[2], o3 = r.k[3], o4 = r.k[4];
    r.k[0] = key < o0 ? key : o0;
    r.k[1] = key < o0 ? o0 : (key < o1 ? key : o1);
    r.k[2] = key < o1 ? o1 : (key < o2 ? key : o2);
    r.k[3] = key < o2 ? o2 : (key < o3 ? key : o3);
    r.k[4] = key < o3 ? o3 : (key < o4 ? key : o4);
}

// G (=8) consecutive lanes share one query.  The block of cells around the query is (2R+1)^2 rows along x
// (R = 2 for 0.5 m cells); lane `sub` owns rows sub, sub + G, … .  A row is skipped when its (y, z) slab is a
// metre or more away, and its x-range is cut to the cells the unit ball reaches in that slab, so the lanes
// scan ~2.3x the ball's volume instead of the 27 m^3 of a 3x3x3 block of 1 m cells.  The cells of a row
// are contiguous in the cell-sorted array: 2 bound loads per row (all in flight together), then the lane's
// rows form one flat candidate list read in batches of KNN_KB (adjacent lanes read different rows, the index
// is L2-resident).  Each lane keeps a private top-5; the G lists are merged by 5 rounds of a group-wide
// (distance, index) minimum.  All G lanes end with the same result.

// r2max: only neighbours with squared distance <= r2max can matter.  KNN_R2_FULL (the largest float below 1) is the
// callers' own gate (sqDis[4] < 1.0); from the second Gauss-Newton iteration on the residual kernel passes the largest
// squared distance from the query's NEW position to its five PREVIOUS neighbours: five map points lie inside that ball, so
// the true fifth-nearest distance cannot exceed it and the scan shrinks from the unit ball to a ~0.4 m one (exact: every
// cell the smaller ball reaches is still scanned, ties included since the test is <=).
constexpr float KNN_R2_FULL = 0x1.fffffep-1f;
// Where a search reads the index from: straight from global memory (L2), or from the wavefront's LDS tile (below).
// start(ok, y, z, x): position, in the accessor's point array, of the first point of cell (x, y, z) — x may be one past the
// row's last cell; pt(i): point i of that array (xyz + map index in the intensity slot).
struct KnnDirect {
    const int* __restrict__ cs; const lvi_pt* __restrict__ pts; int dimx, dimy;
    __device__ __forceinline__ int start(bool ok, int y, int z, int x) const { return cs[ok ? (z * dimy + y) * dimx + x : 0]; }
    __device__ __forceinline__ lvi_pt pt(int i) const { return pts[i]; }
};
template <int G, int KNN_KB, class ACC>
__device__ __forceinline__ void knn5_search_acc(const GridIndex::Meta& m, const ACC& acc, bool act,
                                                float qx, float qy, float qz, int sub, Knn5& out, long long* tk = nullptr, float r2max = KNN_R2_FULL,
                                                bool want_lb = false, float* lb2 = nullptr)
{
    constexpr int KNN_RPL = (25 + G - 1) / G;         // rows per lane
#define LVI_KT(slot) do { if (tk) tk[slot] = clock64(); } while (0)
    LVI_KT(0);
    KnnKeys r;
#pragma unroll
    for (int k = 0; k < 5; k++) r.k[k] = KNN_EMPTY;
    unsigned rej = 0x7F800000u;                       // smallest squared distance (float bits) among the candidates that lose their place
    if (m.ok && m.n > 0) {                            // (wave-uniform)
        const float e = (float)m.edge, inv_e = (float)m.inv_edge;
        // cell of the query (double, as cell_of) and its position inside that cell in [0,1) (f32 is plenty: the row
        // tests below only have to be conservative, and they carry a 1e-4 margin)
        const double gx = ((double)qx - m.origin[0]) * m.inv_edge, gy = ((double)qy - m.origin[1]) * m.inv_edge, gz = ((double)qz - m.origin[2]) * m.inv_edge;
        const double fxd = floor(gx), fyd = floor(gy), fzd = floor(gz);
        const float tx = (float)(gx - fxd), ty = (float)(gy - fyd), tz = (float)(gz - fzd);
        // far outside the grid nothing can be within 1 m (2 cells of padding): clamp so that the ints below cannot overflow
        const int cx = (int)fmin(fmax(fxd, -4.0), (double)m.dim[0] + 4.0), cy = (int)fmin(fmax(fyd, -4.0), (double)m.dim[1] + 4.0),
                  cz = (int)fmin(fmax(fzd, -4.0), (double)m.dim[2] + 4.0);
        // rows (y, z) of the cell block the ball of radius sqrt(r2max) reaches: the bounding rectangle of the ball in cells,
        // never beyond the m.R cells a 1 m ball needs (2 for 0.5 m cells): at most 5 x 5 rows, 3 x 3 or fewer for the ~0.4 m
        // balls of the bounded iterations
        const float rc = sqrtf(r2max + 1e-4f) * inv_e + 1e-4f;
        const int y0 = max((int)floorf(ty - rc), -m.R), y1 = min((int)floorf(ty + rc), m.R);
        const int z0 = max((int)floorf(tz - rc), -m.R), z1 = min((int)floorf(tz + rc), m.R);
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        const int inv_ny = ny == 1 ? 256 : (ny == 2 ? 128 : (ny == 3 ? 86 : (ny == 4 ? 64 : 52)));      // (rr * inv_ny) >> 8 == rr / ny for rr < 25
        int off[KNN_RPL], st[KNN_RPL + 1];
        st[0] = 0;
        // rows a lane of this wavefront can own at most (uniform when the whole wavefront is here; a lane's own need is always
        // covered, so a partly active wavefront is merely less tidy): slots beyond it issue no loads at all
        int tcap = (nrows + G - 1) / G;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tcap = max(tcap, __shfl_xor(tcap, o, 64));
        int rbv[KNN_RPL], rev[KNN_RPL]; bool okv[KNN_RPL];
#pragma unroll
        for (int t = 0; t < KNN_RPL; t++) { rbv[t] = 0; rev[t] = 0; okv[t] = false; }
#pragma unroll
        for (int t = 0; t < KNN_RPL; t++) {
            if (t < tcap) {                                 // wave-uniform: a scalar branch, the loop stays unrolled
                // no branch around the two bound loads: a row that does not count reads cells 0 and 1 and is masked afterwards, so that
                // the loads of all the lane's rows are in flight together (inside nested branches every row waited for its own pair)
                const int rr = sub + G * t;
                const int rq = (rr * inv_ny) >> 8;
                const int dy = y0 + rr - rq * ny, dz = z0 + rq;
                const int y = cy + dy, z = cz + dz;
                bool ok = act && rr < nrows && y >= 0 && y < m.dim[1] && z >= 0 && z < m.dim[2];
                // distance (m) from the query to the row's slab along y and z; 0 inside
                const float ddy = dy == 0 ? 0.f : (dy > 0 ? (float)dy - ty : ty - (float)(dy + 1)) * e;
                const float ddz = dz == 0 ? 0.f : (dz > 0 ? (float)dz - tz : tz - (float)(dz + 1)) * e;
                // margins: the f32 distance of a candidate may round below 1 when the exact one is just above
                const float rem = r2max + 1e-4f - ddy * ddy - ddz * ddz;
                ok = ok && rem > 0.f;
                const float sx = sqrtf(fmaxf(rem, 0.f)) * inv_e + 1e-4f;
                const int x0 = max(cx + (int)floorf(tx - sx), 0), x1 = min(cx + (int)floorf(tx + sx), m.dim[0] - 1);
                ok = ok && x0 <= x1;
                rbv[t] = acc.start(ok, y, z, x0);                   // consumed after the loop: nothing in here waits for a load
                rev[t] = acc.start(ok, y, z, x1 + 1);
                okv[t] = ok;
            }
        }
#pragma unroll
        for (int t = 0; t < KNN_RPL; t++) { off[t] = okv[t] ? rbv[t] : 0; st[t + 1] = okv[t] ? rev[t] - rbv[t] : 0; }
#pragma unroll
        for (int t = 0; t < KNN_RPL; t++) { const int len = st[t + 1]; st[t + 1] = st[t] + len; off[t] -= st[t]; }
        const int T = st[KNN_RPL];
        LVI_KT(1);
        if (tk) tk[5] = T;
        for (int f0 = 0; f0 < T; f0 += KNN_KB) {
            lvi_pt p[KNN_KB];
#pragma unroll
            for (int u = 0; u < KNN_KB; u++) {
                const int f = f0 + u;
                int o = off[0];
#pragma unroll
                for (int t = 1; t < KNN_RPL; t++) o = f >= st[t] ? off[t] : o;
                p[u] = acc.pt(f < T ? f + o : 0);               // unconditional (entry 0 when past the list: masked below): the batch's loads travel together
            }
            LVI_KT(2);
#pragma unroll
            for (int u = 0; u < KNN_KB; u++) {
                const float ex = sub_rn(qx, p[u].x), ey = sub_rn(qy, p[u].y), ez = sub_rn(qz, p[u].z);
                const float dist = add_rn(add_rn(mul_rn(ex, ex), mul_rn(ey, ey)), mul_rn(ez, ez));
                // Only neighbours closer than 1 m can matter: the callers reject a feature unless its 5th neighbour has
                // sqDis < 1.0, and if five such neighbours exist they ARE the five nearest.
                const bool take = f0 + u < T && dist <= r2max;
                const unsigned long long key = take ? knn_key(dist, __float_as_int(p[u].intensity)) : KNN_EMPTY;
                if (want_lb) { const unsigned dr = (unsigned)((key < r.k[4] ? r.k[4] : key) >> 32); rej = dr < rej ? dr : rej; }
                knn_insert(r, key);
            }
        }
    }
    if (tk) tk[4] = 0;
    LVI_KT(3);
    // merge: 5 rounds of a group-wide minimum; the owner of the winner pops it (keys are unique per lane)
#pragma unroll
    for (int k = 0; k < 5; k++) {
        unsigned long long h = r.k[0];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(h, o, 64); h = v < h ? v : h; }
        out.d[k] = __uint_as_float((unsigned)(h >> 32)); out.i[k] = (int)(unsigned)h;
        if (r.k[0] == h && h != KNN_EMPTY) { r.k[0] = r.k[1]; r.k[1] = r.k[2]; r.k[2] = r.k[3]; r.k[3] = r.k[4]; r.k[4] = KNN_EMPTY; }
    }
    if (want_lb) {
        // every map point outside the five is either a candidate that lost its place (in a lane's list or in the merge: the
        // smallest such distance is rej / a lane's remaining head), or was never taken: farther than sqrt(r2max)
        const unsigned hd = (unsigned)(r.k[0] >> 32);
        rej = hd < rej ? hd : rej;
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) { const unsigned v = __shfl_xor(rej, o, 64); rej = v < rej ? v : rej; }
        *lb2 = fminf(__uint_as_float(rej), r2max);
    }
}

template <int G, int KNN_KB>
__device__ __forceinline__ void knn5_search_group(const GridIndex::Meta& m, const int* __restrict__ cell_start, const lvi_pt* __restrict__ sorted,
                                                  float qx, float qy, float qz, int sub, Knn5& out, long long* tk = nullptr, float r2max = KNN_R2_FULL,
                                                  bool want_lb = false, float* lb2 = nullptr)
{
    const KnnDirect acc{cell_start, sorted, m.dim[0], m.dim[1]};
    knn5_search_acc<G, KNN_KB>(m, acc, true, qx, qy, qz, sub, out, tk, r2max, want_lb, lb2);
}

// ---- LDS-staged candidate tiles.  One wavefront searches for 64 / G features that follow each other in the scan's voxel order:
// their balls overlap or lie next to each other, so the wavefront first copies the part of the index all of them can reach —
// the bounding box, in cells, of their search regions: the cell starts of its rows and the points of those rows — into ITS
// region of LDS: two rounds of loads with every lane busy, instead of one dependent round per batch of four candidates and
// lane.  The searches then run on the tile with the same row geometry, hence the same candidates and the same five.  A tile
// that does not fit (features far apart, a dense corner of the map) sends the wavefront down the direct path: same result.
template <int TP, int TC>
struct KnnTile { int cs[TC]; int toff[65]; float4 pt[TP]; };
template <int TP, int TC>
struct KnnTileAcc {
    const KnnTile<TP, TC>* T; int X0, Y0, Z0, nx1, ny;
    __device__ __forceinline__ int start(bool ok, int y, int z, int x) const
    {
        const int r = ok ? (z - Z0) * ny + (y - Y0) : 0;
        const int base = r * nx1;
        return T->cs[base + (ok ? x - X0 : 0)] - T->cs[base] + T->toff[r];
    }
    __device__ __forceinline__ lvi_pt pt(int i) const { const float4 v = T->pt[i]; return lvi_pt{v.x, v.y, v.z, v.w}; }
};
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v; }
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v; }
__device__ __forceinline__ void wave_lds_fence() { __threadfence_block(); __builtin_amdgcn_wave_barrier(); }

// every lane of the wavefront calls this (act: the lane's feature searches).  true: the tile is built and `acc` reads it.
template <int TP, int TC>
__device__ __forceinline__ bool knn_tile_build(const GridIndex::Meta& m, const int* __restrict__ cell_start, const lvi_pt* __restrict__ sorted,
                                               bool act, float qx, float qy, float qz, float r2max, KnnTile<TP, TC>& T, KnnTileAcc<TP, TC>& acc)
{
    if (!(m.ok && m.n > 0)) return false;
    const int lane = lane_id();
    const float inv_e = (float)m.inv_edge;
    // the cells this lane's search can touch (a superset of the rows / x-ranges knn5_search_acc derives from the same values)
    const double gx = ((double)qx - m.origin[0]) * m.inv_edge, gy = ((double)qy - m.origin[1]) * m.inv_edge, gz = ((double)qz - m.origin[2]) * m.inv_edge;
    const double fxd = floor(gx), fyd = floor(gy), fzd = floor(gz);
    const float tx = (float)(gx - fxd), ty = (float)(gy - fyd), tz = (float)(gz - fzd);
    const int cx = (int)fmin(fmax(fxd, -4.0), (double)m.dim[0] + 4.0), cy = (int)fmin(fmax(fyd, -4.0), (double)m.dim[1] + 4.0),
              cz = (int)fmin(fmax(fzd, -4.0), (double)m.dim[2] + 4.0);
    const float rc = sqrtf(r2max + 1e-4f) * inv_e + 2e-4f;
    int xa = max(cx + (int)floorf(tx - rc), 0), xb = min(cx + (int)floorf(tx + rc), m.dim[0] - 1);
    int ya = max(cy + max((int)floorf(ty - rc), -m.R), 0), yb = min(cy + min((int)floorf(ty + rc), m.R), m.dim[1] - 1);
    int za = max(cz + max((int)floorf(tz - rc), -m.R), 0), zb = min(cz + min((int)floorf(tz + rc), m.R), m.dim[2] - 1);
    const bool any = act && xa <= xb && ya <= yb && za <= zb;
    const int BIG = 0x3fffffff;
    const int X0 = wave_min_i(any ? xa : BIG), X1 = wave_max_i(any ? xb : -BIG);
    const int Y0 = wave_min_i(any ? ya : BIG), Y1 = wave_max_i(any ? yb : -BIG);
    const int Z0 = wave_min_i(any ? za : BIG), Z1 = wave_max_i(any ? zb : -BIG);
    acc.T = &T; acc.X0 = X0; acc.Y0 = Y0; acc.Z0 = Z0; acc.nx1 = 1; acc.ny = 1;
    if (X0 > X1) {                                   // nothing to search in reach of the grid: an empty tile serves every lane
        if (lane == 0) { T.cs[0] = 0; T.toff[0] = 0; }
        wave_lds_fence();
        return true;
    }
    const long long nxl = (long long)X1 - X0 + 2, nyl = (long long)Y1 - Y0 + 1, nzl = (long long)Z1 - Z0 + 1;
    if (nyl * nzl > 64 || nxl * nyl * nzl > TC) return false;
    const int nx1 = (int)nxl, ny = (int)nyl, nrows = (int)(nyl * nzl), ncs = nx1 * nrows;
    acc.nx1 = nx1; acc.ny = ny;
    // round 1: the cell starts of the tile's rows, every lane a share, all loads in flight together
    {
        constexpr int NL = (TC + 63) / 64;
        int v[NL];
        const float rcp = 1.0f / (float)nx1;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int e = lane + 64 * k, ec = min(e, ncs - 1);
            int r = (int)((float)ec * rcp);
            r -= (r * nx1 > ec) ? 1 : 0; r += ((r + 1) * nx1 <= ec) ? 1 : 0;
            const int ix = ec - r * nx1;
            const int y = Y0 + r % ny, z = Z0 + r / ny;
            v[k] = (k * 64 < ncs) ? cell_start[(z * m.dim[1] + y) * m.dim[0] + X0 + ix] : 0;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) { const int e = lane + 64 * k; if (e < ncs) T.cs[e] = v[k]; }
    }
    wave_lds_fence();
    // the rows' lengths -> where each row's points sit in the tile
    const int len = lane < nrows ? T.cs[lane * nx1 + nx1 - 1] - T.cs[lane * nx1] : 0;
    const int incl = wave_incl_scan(len);
    const int P = __shfl(incl, 63, 64);
    if (P > TP) return false;
    if (lane < nrows) T.toff[lane] = incl - len;
    if (lane == 0) T.toff[nrows] = P;
    wave_lds_fence();
    // round 2: the points
    {
        constexpr int NL = (TP + 63) / 64;
        lvi_pt v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = lane + 64 * k, ic = min(i, max(P - 1, 0));
            // the row of tile position ic: the last r with toff[r] <= ic (rows are few: a 6-step search in LDS)
            int lo = 0, hi = nrows;                  // toff[lo] <= ic < toff[hi]
#pragma unroll
            for (int it = 0; it < 6; it++) { const int mid = (lo + hi) >> 1; const bool up = mid < hi && T.toff[mid] <= ic; lo = up ? mid : lo; hi = up ? hi : mid; }
            const int src = T.cs[lo * nx1] + (ic - T.toff[lo]);
            v[k] = sorted[(k * 64 < P) ? src : 0];
        }
#pragma unroll
        for (int k = 0; k < NL; k++) { const int i = lane + 64 * k; if (i < P) T.pt[i] = make_float4(v[k].x, v[k].y, v[k].z, v[k].intensity); }
    }
    wave_lds_fence();
    return true;
}

__global__ __launch_bounds__(256) void knn_debug_kernel(const GridIndex::Meta* meta, const int* cell_start, const lvi_pt* sorted,
                                                        const lvi_pt* q, int nq, int* idx, float* sqd)
{
    // same group search the residual kernels use
    const int t = blockIdx.x * (256 / KNN_G) + threadIdx.x / KNN_G, sub = threadIdx.x % KNN_G;
    if (t >= nq) return;
    Knn5 r;
    knn5_search_group<KNN_G, 8>(*meta, cell_start, sorted, q[t].x, q[t].y, q[t].z, sub, r);
    if (sub != 0) return;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const bool ok = r.d[k] < 1.0f;
        idx[t * 5 + k] = ok ? r.i[k] : -1;
        sqd[t * 5 + k] = ok ? r.d[k] : INFINITY;
    }
}

// ------------------------------------------------------------------------------------------- small matrices
// symmetric 3x3 eigen-decomposition, cyclic Jacobi in f32: eigenvalues descending, v0 = eigenvector of the largest
__device__ void eig3_sym(float a11, float a12, float a13, float a22, float a23, float a33, float ev[3], float v0[3])
{
    float A[3][3] = {{a11, a12, a13}, {a12, a22, a23}, {a13, a23, a33}};
    float V[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
    for (int sweep = 0; sweep < 12; sweep++) {
        const float off = fabsf(A[0][1]) + fabsf(A[0][2]) + fabsf(A[1][2]);
        const float dia = fabsf(A[0][0]) + fabsf(A[1][1]) + fabsf(A[2][2]);
        if (off <= 1e-12f * dia || off == 0.f) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            const float apq = A[p][q];
            if (apq == 0.f) continue;
            const float theta = (A[q][q] - A[p][p]) / (2.f * apq);
            const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
            const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
            A[p][p] -= t * apq; A[q][q] += t * apq; A[p][q] = A[q][p] = 0.f;
            const int r = 3 - p - q;
            const float arp = A[r][p], arq = A[r][q];
            A[r][p] = A[p][r] = c * arp - s * arq;
            A[r][q] = A[q][r] = s * arp + c * arq;
#pragma unroll
            for (int k = 0; k < 3; k++) { const float vp = V[k][p], vq = V[k][q]; V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq; }
        }
    }
    const float l0 = A[0][0], l1 = A[1][1], l2 = A[2][2];
    const int i0 = (l0 >= l1 && l0 >= l2) ? 0 : (l1 >= l2 ? 1 : 2);
    ev[0] = i0 == 0 ? l0 : (i0 == 1 ? l1 : l2);
    ev[1] = i0 == 0 ? fmaxf(l1, l2) : (i0 == 1 ? fmaxf(l0, l2) : fmaxf(l0, l1));
    ev[2] = i0 == 0 ? fminf(l1, l2) : (i0 == 1 ? fminf(l0, l2) : fminf(l0, l1));
#pragma unroll
    for (int k = 0; k < 3; k++) v0[k] = i0 == 0 ? V[k][0] : (i0 == 1 ? V[k][1] : V[k][2]);
}

// least squares of the 5x3 system A x = b by Householder QR with column pivoting (f32).
// Column-major registers and compile-time indices only (a runtime-indexed private array is demoted
// to LDS/scratch by hipcc and made this the slowest part of the residual kernel).
__device__ __forceinline__ void lstsq_5x3(float A[5][3], float b[5], float x[3])
{
    float c[3][5];                       // c[col][row]
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 5; i++) c[j][i] = A[i][j];
    int p0 = 0, p1 = 1, p2 = 2;          // perm: column k of the factorisation is original column p_k
    int rank = 3;
    float thr0 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) { float s = 0.f;
#pragma unroll
        for (int i = 0; i < 5; i++) s += c[j][i] * c[j][i];
        thr0 = fmaxf(thr0, sqrtf(s)); }
    const float thr = (thr0 * 1.1920929e-7f) * (thr0 * 1.1920929e-7f) / 5.f;
    float diag[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // largest remaining column norm (rows k..4)
        float nrm[3] = {-1.f, -1.f, -1.f};
#pragma unroll
        for (int j = 0; j < 3; j++) if (j >= k) { float s = 0.f;
#pragma unroll
            for (int i = 0; i < 5; i++) if (i >= k) s += c[j][i] * c[j][i];
            nrm[j] = s; }
        int big = k; float bigsq = nrm[k];
#pragma unroll
        for (int j = 0; j < 3; j++) if (j > k && nrm[j] > bigsq) { bigsq = nrm[j]; big = j; }
        if (rank == 3 && bigsq < thr * (float)(5 - k)) rank = k;
#pragma unroll
        for (int j = 0; j < 3; j++) if (j > k && big == j) {
#pragma unroll
            for (int i = 0; i < 5; i++) { const float t = c[k][i]; c[k][i] = c[j][i]; c[j][i] = t; }
            // swap perm[k] and perm[j]
            int pk = (k == 0) ? p0 : (k == 1 ? p1 : p2), pj = (j == 1) ? p1 : p2;
            if (k == 0) p0 = pj; else if (k == 1) p1 = pj;
            if (j == 1) p1 = pk; else p2 = pk;
        }
        const float c0 = c[k][k];
        float tail = 0.f;
#pragma unroll
        for (int i = 0; i < 5; i++) if (i > k) tail += c[k][i] * c[k][i];
        float tau, beta;
        if (tail <= 1.17549435e-38f) {
            tau = 0.f; beta = c0;
#pragma unroll
            for (int i = 0; i < 5; i++) if (i > k) c[k][i] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tail); if (c0 >= 0.f) beta = -beta;
#pragma unroll
            for (int i = 0; i < 5; i++) if (i > k) c[k][i] /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        diag[k] = beta;
#pragma unroll
        for (int j = 0; j < 3; j++) if (j > k) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 5; i++) if (i > k) s += c[k][i] * c[j][i];
            s += c[j][k]; s *= tau; c[j][k] -= s;
#pragma unroll
            for (int i = 0; i < 5; i++) if (i > k) c[j][i] -= s * c[k][i];
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 5; i++) if (i > k) s += c[k][i] * b[i];
        s += b[k]; s *= tau; b[k] -= s;
#pragma unroll
        for (int i = 0; i < 5; i++) if (i > k) b[i] -= s * c[k][i];
    }
    // back substitution on the leading rank x rank block: R(i,j) = c[j][i] for j > i, R(i,i) = diag[i]
    float y0 = 0.f, y1 = 0.f, y2 = 0.f;
    if (rank >= 3) y2 = b[2] / diag[2];
    if (rank >= 2) y1 = (b[1] - (rank >= 3 ? c[2][1] * y2 : 0.f)) / diag[1];
    if (rank >= 1) y0 = (b[0] - (rank >= 2 ? c[1][0] * y1 : 0.f) - (rank >= 3 ? c[2][0] * y2 : 0.f)) / diag[0];
    x[0] = x[1] = x[2] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float v = (p0 == j ? y0 : 0.f) + (p1 == j ? y1 : 0.f) + (p2 == j ? y2 : 0.f);
        x[j] = v;
    }
}

// ------------------------------------------------------------------------------------------- residuals
struct IcpArgs {
    IcpState* st;
    const lvi_pt* q[2]; const int* nq;            // cornerDS / surfDS, voxScan.d_nout
    const GridIndex::Meta* meta[2]; const int* cell_start[2]; const lvi_pt* sorted[2];
    const lvi_pt* mapds[2];
    int edgeMin, surfMin, max_iters, disable_break;
    float rot_tol, z_tol; double imu_weight;
    int imu_available; float imu_roll, imu_pitch;
    void* d_record;
    long long* cyc;
    const int* d_status;                          // [2] device error words: scan side, map build
    int have_map;
    // per-feature records of the Gauss-Newton loop, structure of arrays with stride `cap` (one lane per feature reads them coalesced)
    int cap;
    int* nn_prev;                                 // [5][cap] the five neighbours of the feature's last search, in the (distance, index) order its FIT was made for
                                                  //          ([0][t] = -1: fewer than five within 1 m); nullptr: every iteration searches the whole unit ball
    float4* nn_pt;                                // [5][cap] their coordinates (xyz), same order: the skip test, the order test and the fit read no map point
    float4* nn_ref;                               // [cap] where the feature stood at its last search (xyz) and a lower bound (w, squared) on the distance from there
                                                  //       to every map point outside its five; nullptr: search in every iteration
    float4* fit;                                  // [cap] surf: the plane (pa, pb, pc, pd); corner: the first line point
    float4* fit2;                                 // [cap] corner: the second line point
    unsigned char* fit_ok;                        // [cap] 0: no fit stored; 1: stored, geometric gate passed; 2: stored, gate failed
    float knn_slack;                              // metres added to the radius of a bounded search (room for later iterations to skip theirs)
    int lds_tiles;                                // LVI_KNN_TILES=1: phase A stages the index tile of a wavefront's features in LDS when it fits (4 lanes per feature; same bits; see DESIGN for why it is not the default)
    IcpHostResult* h_res; const int* nout3;       // pinned host block the finish step fills (lvi_scan_match reads it after one wait); voxScan.d_nout
    int* h_feat;                                  // pinned host word: this match's feature count, for the next launches' grid (LidarDev::h_gn_feat)
    int stamp_iter;                               // the iteration whose phase stamps are kept in cyc[] (LVI_ICP_STAMP_ITER, default: the last one launched)
    int xcd_map;                                  // residual workgroups are dealt to the XCDs in contiguous feature ranges (LVI_ICP_NO_XCD_MAP=1: in launch order)
    // normal equations: 28 columns x {coarse, fine} exact fixed-point accumulators, ICP_SHARDS shards (workgroup & 7), three
    // buffers in rotation: launch i adds into buffer i % 3, reads the totals of launch i - 1 from (i - 1) % 3 and zeroes (i + 1) % 3
    unsigned long long* acc;                      // [3][ICP_SHARDS][56]
};
constexpr int ICP_SHARDS = 8;

__device__ __forceinline__ lvi_pt to_map(const float A[12], const lvi_pt& p)       // pointAssociateToMap :339-345
{
    lvi_pt o;
    o.x = A[0] * p.x + A[1] * p.y + A[2] * p.z + A[3];
    o.y = A[4] * p.x + A[5] * p.y + A[6] * p.z + A[7];
    o.z = A[8] * p.x + A[9] * p.y + A[10] * p.z + A[11];
    o.intensity = p.intensity;
    return o;
}

// cornerOptimization (:1006-1096) in two halves.  The FIT depends on the five neighbours alone (their coordinates and their
// order): centroid, covariance, cv::eigen, the eigenvalue gate and the two points on the line.  The EVALUATION takes the
// feature's current position.  A Gauss-Newton iteration whose five neighbours and their order did not change re-uses the fit
// of the previous one: the same bits as fitting again.
__device__ __forceinline__ bool corner_fit(const lvi_pt nb[5], float4& l1, float4& l2)
{
    float cx = 0, cy = 0, cz = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { cx += nb[j].x; cy += nb[j].y; cz += nb[j].z; }
    cx /= 5; cy /= 5; cz /= 5;
    float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const float ax = nb[j].x - cx, ay = nb[j].y - cy, az = nb[j].z - cz;
        a11 += ax * ax; a12 += ax * ay; a13 += ax * az; a22 += ay * ay; a23 += ay * az; a33 += az * az;
    }
    a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
    float ev[3], v0[3];
    eig3_sym(a11, a12, a13, a22, a23, a33, ev, v0);                                // cv::eigen :1050
    const float x1 = cx + 0.1 * v0[0], y1 = cy + 0.1 * v0[1], z1 = cz + 0.1 * v0[2];
    const float x2 = cx - 0.1 * v0[0], y2 = cy - 0.1 * v0[1], z2 = cz - 0.1 * v0[2];
    l1 = make_float4(x1, y1, z1, 0.f); l2 = make_float4(x2, y2, z2, 0.f);
    return ev[0] > 3 * ev[1];
}
__device__ __forceinline__ bool corner_eval(const float4& l1, const float4& l2, const lvi_pt& pointSel, lvi_pt& coeff)
{
    const float x0 = pointSel.x, y0 = pointSel.y, z0 = pointSel.z;
    const float x1 = l1.x, y1 = l1.y, z1 = l1.z, x2 = l2.x, y2 = l2.y, z2 = l2.z;
    const float m11 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
    const float m12 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
    const float m13 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
    const float a012 = sqrtf(m11 * m11 + m12 * m12 + m13 * m13);
    const float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
    const float la = ((y1 - y2) * m11 + (z1 - z2) * m12) / a012 / l12;
    const float lb = -((x1 - x2) * m11 - (z1 - z2) * m13) / a012 / l12;
    const float lc = -((x1 - x2) * m12 + (y1 - y2) * m13) / a012 / l12;
    const float ld2 = a012 / l12;
    const float s = 1 - 0.9 * fabsf(ld2);
    coeff.x = s * la; coeff.y = s * lb; coeff.z = s * lc; coeff.intensity = s * ld2;
    return s > 0.1;
}
__device__ bool corner_residual(const IcpArgs& a, const lvi_pt& pointSel, const Knn5& r, lvi_pt& coeff)
{
    if (!(r.d[4] < 1.0f)) return false;                                            // :1025
    const lvi_pt* map = a.mapds[0];
    lvi_pt nb[5];
#pragma unroll
    for (int j = 0; j < 5; j++) nb[j] = map[r.i[j]];
    float4 l1, l2;
    if (!corner_fit(nb, l1, l2)) return false;
    return corner_eval(l1, l2, pointSel, coeff);
}

// surfOptimization (:1098-1167), the same two halves: the plane through the five neighbours and its 0.2 m gate / the feature's
// distance from it
__device__ __forceinline__ bool surf_fit(const lvi_pt nb[5], float4& pl)
{
    float M[5][3], b[5], X[3];
#pragma unroll
    for (int j = 0; j < 5; j++) { M[j][0] = nb[j].x; M[j][1] = nb[j].y; M[j][2] = nb[j].z; b[j] = -1.f; }
    lstsq_5x3(M, b, X);                                                            // colPivHouseholderQr().solve :1128
    float pa = X[0], pb = X[1], pc = X[2], pd = 1;
    const float ps = sqrtf(pa * pa + pb * pb + pc * pc);
    pa /= ps; pb /= ps; pc /= ps; pd /= ps;
    bool valid = true;
#pragma unroll
    for (int j = 0; j < 5; j++)
        if (fabsf(pa * nb[j].x + pb * nb[j].y + pc * nb[j].z + pd) > 0.2) valid = false;
    pl = make_float4(pa, pb, pc, pd);
    return valid;
}
__device__ __forceinline__ bool surf_eval(const float4& pl, const lvi_pt& pointOri, const lvi_pt& pointSel, lvi_pt& coeff)
{
    const float pa = pl.x, pb = pl.y, pc = pl.z, pd = pl.w;
    const float pd2 = pa * pointSel.x + pb * pointSel.y + pc * pointSel.z + pd;
    const float s = 1 - 0.9 * fabsf(pd2) / sqrtf(sqrtf(pointOri.x * pointOri.x + pointOri.y * pointOri.y + pointOri.z * pointOri.z));
    coeff.x = s * pa; coeff.y = s * pb; coeff.z = s * pc; coeff.intensity = s * pd2;
    return s > 0.1;
}
__device__ bool surf_residual(const IcpArgs& a, const lvi_pt& pointOri, const lvi_pt& pointSel, const Knn5& r, lvi_pt& coeff)
{
    if (!(r.d[4] < 1.0f)) return false;                                            // :1121
    const lvi_pt* map = a.mapds[1];
    lvi_pt nb[5];
#pragma unroll
    for (int j = 0; j < 5; j++) nb[j] = map[r.i[j]];
    float4 pl;
    if (!surf_fit(nb, pl)) return false;
    return surf_eval(pl, pointOri, pointSel, coeff);
}

// one Gauss-Newton row: matA(i, 0..5), matB(i) (LMOptimization :1224-1255)
__device__ __forceinline__ void lm_row(const float tr[6], const lvi_pt& ori, const lvi_pt& cf, float rowA[6], float& rowB)
{
    const float srx = tr[0], crx = tr[1], sry = tr[2], cry = tr[3], srz = tr[4], crz = tr[5];
    const float px = ori.y, py = ori.z, pz = ori.x;          // lidar -> camera
    const float cx = cf.y, cy = cf.z, cz = cf.x;
    const float arx = (crx * sry * srz * px + crx * crz * sry * py - srx * sry * pz) * cx
                    + (-srx * srz * px - crz * srx * py - crx * pz) * cy
                    + (crx * cry * srz * px + crx * cry * crz * py - cry * srx * pz) * cz;
    const float ary = ((cry * srx * srz - crz * sry) * px + (sry * srz + cry * crz * srx) * py + crx * cry * pz) * cx
                    + ((-cry * crz - srx * sry * srz) * px + (cry * srz - crz * srx * sry) * py - crx * sry * pz) * cz;
    const float arz = ((crz * srx * sry - cry * srz) * px + (-cry * crz - srx * sry * srz) * py) * cx
                    + (crx * crz * px - crx * srz * py) * cy
                    + ((sry * srz + cry * crz * srx) * px + (crz * sry - cry * srx * srz) * py) * cz;
    rowA[0] = arz; rowA[1] = arx; rowA[2] = ary; rowA[3] = cz; rowA[4] = cx; rowA[5] = cy;
    rowB = -cf.intensity;
}

__device__ void make_pose(IcpPose& p)
{
    // pcl::getTransformation(x,y,z,roll,pitch,yaw) (trans2Affine3f :404-407)
    // sinf / cosf of the host (glibc: correctly rounded but for rare cases) are matched by rounding the double result;
    // the device's own float versions are 1-2 ulp off, enough to move a point across a selection threshold
    double sr, cr, sp, cp, sy, cy;
    sincos((double)p.T[0], &sr, &cr); sincos((double)p.T[1], &sp, &cp); sincos((double)p.T[2], &sy, &cy);
    const float A = (float)cy, B = (float)sy, C = (float)cp, D = (float)sp, E = (float)cr, F = (float)sr, DE = D * E, DF = D * F;
    p.A[0] = A * C; p.A[1] = A * DF - B * E; p.A[2] = B * F + A * DE; p.A[3] = p.T[3];
    p.A[4] = B * C; p.A[5] = A * E + B * DF; p.A[6] = B * DE - A * F; p.A[7] = p.T[4];
    p.A[8] = -D;    p.A[9] = C * F;          p.A[10] = C * E;         p.A[11] = p.T[5];
    // LMOptimization :1202-1207 — srx from pitch, sry from yaw, srz from roll
    p.trig[0] = D; p.trig[1] = C;
    p.trig[2] = B; p.trig[3] = A;
    p.trig[4] = F; p.trig[5] = E;
}

// The initial guess of a scan match and the reset of its Gauss-Newton state (what scan2MapOptimization starts from, :1315-1322):
// one thread per batch slot.  The feature-count gates (:1317, :1320) are evaluated by the first Gauss-Newton launch, where the
// counts of the scan's grids exist.
struct PoseInitArgs { float* dst; float t[6]; int* d_status; IcpState* st; long long* cyc; unsigned long long* acc; };
__global__ __launch_bounds__(64) void set_pose_init_kernel(Batch<PoseInitArgs> B_)
{
    const PoseInitArgs& a = B_.a[blockIdx.z];
    for (int k = threadIdx.x; k < ICP_SHARDS * 56; k += 64) a.acc[k] = 0ull;      // buffer 0: the first launch adds into it
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int k = 0; k < 6; k++) a.dst[k] = a.t[k];
    if (a.d_status) a.d_status[0] = 0;            // scan-side device status word of a batch slot (single scans: cleared by the upload)
    IcpState& s = *a.st;
    for (int k = 0; k < 6; k++) s.pose[0].T[k] = a.t[k];
    make_pose(s.pose[0]);
    s.cur = 0;
    s.done = 0; s.converged = 0; s.degenerate = 0; s.iters = 0; s.any_lm = 0; s.status = LVI_OK;
    a.cyc[15] = 0; a.cyc[10] = 0; a.cyc[11] = 0;   // searches / wavefront rounds (direct, tiled) counted by the Gauss-Newton kernel (debug read-out)
    for (int i = 0; i < LVI_ICP_MAX_ITERS; i++) s.n_sel[i] = 0;
}

// tf2 pieces of transformUpdate (doubles)
struct Quatd { double x, y, z, w; };
__device__ Quatd q_rpy(double roll, double pitch, double yaw)
{
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    return Quatd{sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
}
__device__ double q_dot(const Quatd& a, const Quatd& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ Quatd q_slerp(const Quatd& a, const Quatd& q, double t)
{
    const double s = sqrt(q_dot(a, a) * q_dot(q, q));
    const double d = q_dot(a, q);
    const double theta = ((d < 0) ? acos(-d / s) * 2.0 : acos(d / s) * 2.0) / 2.0;
    if (theta != 0.0) {
        const double dd = 1.0 / sin(theta), s0 = sin((1.0 - t) * theta), s1 = sin(t * theta);
        const double sg = (d < 0) ? -1.0 : 1.0;
        return Quatd{(a.x * s0 + sg * q.x * s1) * dd, (a.y * s0 + sg * q.y * s1) * dd, (a.z * s0 + sg * q.z * s1) * dd, (a.w * s0 + sg * q.w * s1) * dd};
    }
    return a;
}
__device__ void q_to_rpy(const Quatd& q, double& roll, double& pitch, double& yaw)
{
    const double s = 2.0 / q_dot(q, q);
    const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (fabs(m20) >= 1) { yaw = 0; roll = atan2(m21, m22); pitch = (m20 < 0) ? M_PI / 2.0 : -M_PI / 2.0; }
    else { pitch = -asin(m20); roll = atan2(m21 / cos(pitch), m22 / cos(pitch)); yaw = atan2(m10 / cos(pitch), m00 / cos(pitch)); }
}

// transformUpdate (:1345-1375) and the 32-byte pose record of the scan: one lane, at the end of the last launch
__device__ void icp_finish_body(const IcpArgs& a)
{
    IcpState& s = *a.st;
    // (works on copies: the loop state stays as it is, so that a caller that enqueues the loop in chunks — lvi_scan_match with
    // the reference's break rule — can go on after looking at this record)
    float T[6];
    for (int k = 0; k < 6; k++) T[k] = s.pose[s.cur & 1].T[k];
    int status = s.status;
    const bool ran = (status == LVI_OK);
    if (ran) {
        for (int k = 0; k < 6; k++) s.pose_trace[s.iters * 6 + k] = T[k];
        if (a.imu_available && fabsf(a.imu_pitch) < 1.4f) {                       // transformUpdate :1347-1367
            double r, p, y;
            q_to_rpy(q_slerp(q_rpy(T[0], 0, 0), q_rpy(a.imu_roll, 0, 0), a.imu_weight), r, p, y);
            T[0] = (float)r;
            q_to_rpy(q_slerp(q_rpy(0, T[1], 0), q_rpy(0, a.imu_pitch, 0), a.imu_weight), r, p, y);
            T[1] = (float)p;
        }
        T[0] = fminf(fmaxf(T[0], -a.rot_tol), a.rot_tol);                         // :1370-1372
        T[1] = fminf(fmaxf(T[1], -a.rot_tol), a.rot_tol);
        T[5] = fminf(fmaxf(T[5], -a.z_tol), a.z_tol);
        if (!s.any_lm) status = LVI_TOO_FEW_CORRESPONDENCES;
    }
    // a device-side error of the stages that fed this scan match (sector capacity, KNN grid size) travels in the record:
    // the async / replay entry points have no other channel back to the caller
    const int dev = a.d_status[0] | a.d_status[1];
    if (dev) status = (dev & DEV_ERR_SECTOR_HANDOVER) ? LVI_ERR_HIP : LVI_ERR_CAPACITY;
    for (int k = 0; k < 6; k++) { s.final_pose[k] = T[k]; s.record.pose[k] = T[k]; }
    s.final_status = status;
    s.record.status = status; s.record.iters = s.iters;
    if (a.d_record) *reinterpret_cast<lvi_pose_record*>(a.d_record) = s.record;
}
// ------------------------------------------------------------------------------------------- the 6 x 6 end of an iteration, on ONE wavefront
// Lane j < 6 holds column j of the symmetric 6 x 6 matrix, lane 6 the right-hand side; values that every lane needs are read
// across with v_readlane (compile-time lane and register): a few dozen registers instead of the ~250 of the all-in-one-lane form,
// which is what lets the solve live at the end of the Gauss-Newton kernel.  Control flow is wave-uniform throughout.
__device__ __forceinline__ float rdl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// Householder QR solve of the 6x6 system (cv::solve DECOMP_QR, :1260), f32: c[i] = A[i][lane] for lane < 6, = b[i] on lane 6.
// Returns false when the matrix is singular to working precision; X (every lane) = the solution.  Same operations in the
// same order as the column-by-column scalar form.
__device__ __forceinline__ bool solve6_qr_wave(float c[6], int lane, float X[6])
{
    bool ok = true;
#pragma unroll
    for (int l = 0; l < 6; l++) {
        float vl[6];
        float nrm = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) if (i >= l) { vl[i] = rdl(c[i], l); nrm += vl[i] * vl[i]; }
        const float t0 = vl[l];
        vl[l] = vl[l] + (vl[l] >= 0.f ? 1.f : -1.f) * sqrtf(nrm);
        nrm = sqrtf(nrm + vl[l] * vl[l] - t0 * t0);
        if (nrm == 0.f) { ok = false; nrm = 1.f; }
#pragma unroll
        for (int i = 0; i < 6; i++) if (i >= l) vl[i] /= nrm;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) if (i >= l) s += vl[i] * c[i];
        if (lane >= l) {
#pragma unroll
            for (int i = 0; i < 6; i++) if (i >= l) c[i] -= 2 * vl[i] * s;
        }
    }
    float b[6];
#pragma unroll
    for (int i = 0; i < 6; i++) b[i] = rdl(c[i], 6);
#pragma unroll
    for (int i = 5; i >= 0; i--) {
#pragma unroll
        for (int j = 5; j >= 0; j--) if (j > i) b[i] -= b[j] * rdl(c[i], j);
        const float d = rdl(c[i], i);
        if (fabsf(d) < 1.1920929e-6f) ok = false;
        b[i] /= d;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) X[i] = b[i];
    return ok;
}

// Is every eigenvalue of the symmetric 6 x 6 matrix (the f32 AtA, upper triangle in sums[0..20]) above sigma?  LDL^T of
// A - sigma I in double: all pivots positive <=> yes.  Uniform scalar code (every lane computes the same).
__device__ __forceinline__ bool all_eig_above(const double* sums, double sigma)
{
    double L[6][6], d[6];
    bool pd = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double dj = (double)(float)sums[j * 6 - (j * (j - 1)) / 2] - sigma;
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < j) dj -= L[j][k] * L[j][k] * d[k];
        d[j] = dj;
        if (!(dj > 0.0)) { pd = false; d[j] = 1.0; }
#pragma unroll
        for (int i = 0; i < 6; i++) if (i > j) {
            double v = (double)(float)sums[j * 6 - (j * (j - 1)) / 2 + (i - j)];
#pragma unroll
            for (int k = 0; k < 6; k++) if (k < j) v -= L[i][k] * L[j][k] * d[k];
            L[i][j] = v / d[j];
        }
    }
    return pd;
}

// Symmetric 6x6 eigen-decomposition (cv::eigen :1268), cyclic Jacobi f32 on one wavefront: ac[i] = A[i][lane], vc[i] = V[i][lane]
// (rows of V = eigenvectors).  On return W (every lane) holds the eigenvalues, descending, the rows of V follow.
__device__ __forceinline__ void eig6_wave(float ac[6], float vc[6], int lane, float W[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) vc[i] = (i == lane) ? 1.f : 0.f;
    for (int sweep = 0; sweep < 30; sweep++) {
        float off_l = 0.f, dia_l = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) { off_l += (i < lane) ? fabsf(ac[i]) : 0.f; dia_l += (i == lane) ? fabsf(ac[i]) : 0.f; }
        float off = 0.f, dia = 0.f;
#pragma unroll
        for (int j = 0; j < 6; j++) { off += rdl(off_l, j); dia += rdl(dia_l, j); }
        if (off <= 1e-10f * dia || off == 0.f) break;
#pragma unroll
        for (int p = 0; p < 5; p++)
#pragma unroll
            for (int q = p + 1; q < 6; q++) {
                const float apq = rdl(ac[p], q);
                if (apq == 0.f) continue;
                const float app = rdl(ac[p], p), aqq = rdl(ac[q], q);
                const float theta = (aqq - app) / (2.f * apq);
                const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
                const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
                // lanes r != p, q: rows p and q of their column (A[p][r] = A[r][p] by symmetry)
                const float arp = ac[p], arq = ac[q];
                const float np = c * arp - s * arq, nq = s * arp + c * arq;
                if (lane != p && lane != q) { ac[p] = np; ac[q] = nq; }
                // columns p and q receive the new A[r][p], A[r][q]
#pragma unroll
                for (int r = 0; r < 6; r++) if (r != p && r != q) {
                    const float vp_ = rdl(ac[p], r), vq_ = rdl(ac[q], r);
                    if (lane == p) ac[r] = vp_;
                    if (lane == q) ac[r] = vq_;
                }
                if (lane == p) { ac[p] = app - t * apq; ac[q] = 0.f; }
                if (lane == q) { ac[q] = aqq + t * apq; ac[p] = 0.f; }
                const float vp = vc[p], vq = vc[q];
                vc[p] = c * vp - s * vq; vc[q] = s * vp + c * vq;
            }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) W[i] = rdl(ac[i], i);
    // descending order, rows of V follow their eigenvalue (W is the same in every lane: uniform branches)
#pragma unroll
    for (int k = 0; k < 5; k++) {
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            if (W[k] < W[i]) {
                const float t = W[i]; W[i] = W[k]; W[k] = t;
                const float u = vc[i]; vc[i] = vc[k]; vc[k] = u;
            }
        }
    }
}

// pcl::getTransformation + the trigonometric terms of LMOptimization from T (every lane the same T): three lanes take one
// angle each (sincos in double, as make_pose), every lane assembles the matrix from their results
__device__ __forceinline__ void make_pose_wave(IcpPose& p, int lane)
{
    double sn, cs;
    const float ang = lane == 0 ? p.T[0] : (lane == 1 ? p.T[1] : p.T[2]);
    sincos((double)ang, &sn, &cs);                     // (every lane: lanes >= 3 repeat lane 2's angle)
    const float fs = (float)sn, fc = (float)cs;
    const float F = rdl(fs, 0), E = rdl(fc, 0), D = rdl(fs, 1), C = rdl(fc, 1), B = rdl(fs, 2), A = rdl(fc, 2), DE = D * E, DF = D * F;
    p.A[0] = A * C; p.A[1] = A * DF - B * E; p.A[2] = B * F + A * DE; p.A[3] = p.T[3];
    p.A[4] = B * C; p.A[5] = A * E + B * DF; p.A[6] = B * DE - A * F; p.A[7] = p.T[4];
    p.A[8] = -D;    p.A[9] = C * F;          p.A[10] = C * E;         p.A[11] = p.T[5];
    p.trig[0] = D; p.trig[1] = C; p.trig[2] = B; p.trig[3] = A; p.trig[4] = F; p.trig[5] = E;
}

// The end of Gauss-Newton iteration `iter` — combineOptimizationCoeffs + LMOptimization past the row products (:1169-1313) —
// on one wavefront: sums[28] (LDS) = the 21 + 6 sums of AtA / AtB and the number of selected rows, P.T = the pose the rows were
// made with.  On return P is the pose of iteration iter + 1; the return value says whether the loop is over (converged and
// break enabled).  EVERY workgroup of the next launch runs this on the same integers, so every workgroup holds the same pose
// without any exchange inside a launch; the one with `writer` set also keeps the scan's state in global memory.
__device__ __forceinline__ int icp_iter_end(const IcpArgs& a, int iter, const double* sums, int lane, bool writer, int degen_in, IcpPose& P)
{
    IcpState& s = *a.st;
    const int nsel = (int)sums[27];
    int done = 0;
    if (writer) {
        if (lane == 0) { s.n_sel[iter] = nsel; s.iters = iter + 1; }
        if (lane < 6) s.pose_trace[iter * 6 + lane] = lane == 0 ? P.T[0] : lane == 1 ? P.T[1] : lane == 2 ? P.T[2] : lane == 3 ? P.T[3] : lane == 4 ? P.T[4] : P.T[5];
        if (lane < 27) s.jtj[iter * 27 + lane] = (float)sums[lane];
    }
    if (nsel >= 50) {                                                            // :1210 (uniform)
        // column `lane` of AtA (lanes 0..5), AtB on lane 6
        float c[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int r = min(i, min(lane, 5)), cc = max(i, min(lane, 5));
            const int k = r * 6 - (r * (r - 1)) / 2 + (cc - r);
            c[i] = lane == 6 ? (float)sums[21 + i] : (float)sums[k];
        }
        float ac[6];
#pragma unroll
        for (int i = 0; i < 6; i++) ac[i] = c[i];
        float X[6];
        if (!solve6_qr_wave(c, lane, X)) {
#pragma unroll
            for (int r = 0; r < 6; r++) X[r] = 0.f;
        }
        int degenerate = iter == 0 ? 0 : degen_in;                                // (written by the launch that ended iteration 0, read at this launch's entry)
        float mp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                              // column `lane` of matP; the local matP shadows the member: zero after iteration 0
        if (iter == 0) {
            // isDegenerate <=> an eigenvalue of AtA below 100 (:1262-1291).  A factorisation of AtA - sigma I settles the
            // common case: with every eigenvalue above 100 + 1e-4 trace (the Jacobi iteration's own error is two orders
            // below that margin) nothing of cv::eigen's output is used, and the iteration does not run at all.
            double tr = 0.0;
#pragma unroll
            for (int r = 0; r < 6; r++) tr += fabs((double)(float)sums[r * 6 - (r * (r - 1)) / 2]);
            if (!all_eig_above(sums, 100.0 + 1e-4 * tr)) {
                float vc[6], W[6];
                eig6_wave(ac, vc, lane, W);
                float v2[6];
#pragma unroll
                for (int i = 0; i < 6; i++) v2[i] = vc[i];
                bool below = true;
#pragma unroll
                for (int i = 5; i >= 0; i--) {
                    below = below && (W[i] < 100.f);
                    if (below) { v2[i] = 0.f; degenerate = 1; }
                }
                // matP = matV.inv() * matV2; V is orthogonal, inv(V) = V^T: matP[r][lane] = sum_k V[k][r] V2[k][lane]
#pragma unroll
                for (int r = 0; r < 6; r++) { double acc = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) acc += (double)rdl(vc[k], r) * (double)v2[k];
                    mp[r] = (float)acc; }
            }
        }
        if (degenerate) {
            float X2[6];
#pragma unroll
            for (int r = 0; r < 6; r++) X2[r] = X[r];
#pragma unroll
            for (int r = 0; r < 6; r++) { double acc = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) acc += (double)rdl(mp[r], k) * (double)X2[k];
                X[r] = (float)acc; }
        }
#pragma unroll
        for (int r = 0; r < 6; r++) P.T[r] += X[r];
        // pow(x, 2) of the reference is the correctly rounded double square
        const double r0 = (double)(X[0] * 57.29578f), r1 = (double)(X[1] * 57.29578f), r2 = (double)(X[2] * 57.29578f);
        const double u0 = (double)(X[3] * 100), u1 = (double)(X[4] * 100), u2 = (double)(X[5] * 100);
        const double dR = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
        const double dT = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
        const float deltaR = (float)dR, deltaT = (float)dT;
        const bool conv = deltaR < 0.05 && deltaT < 0.05;                         // :1309
        if (conv && !a.disable_break) done = 1;
        if (writer && lane == 0) {
            s.any_lm = 1;
            if (iter == 0) s.degenerate = degenerate;
            if (conv) { s.converged = 1; if (!a.disable_break) s.done = 1; }
        }
    }
    make_pose_wave(P, lane);
    if (writer) {
        IcpPose& out = s.pose[(iter + 1) & 1];
        if (lane < 6) out.T[lane] = lane == 0 ? P.T[0] : lane == 1 ? P.T[1] : lane == 2 ? P.T[2] : lane == 3 ? P.T[3] : lane == 4 ? P.T[4] : P.T[5];
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) out.A[k] = P.A[k];
#pragma unroll
            for (int k = 0; k < 6; k++) out.trig[k] = P.trig[k];
            s.cur = (iter + 1) & 1;
        }
    }
    return done;
}

// the totals of a launch: 8 shards x 56 exact integers -> sums[28] (one wavefront; the shards' loads travel together)
__device__ __forceinline__ void icp_take_sums(const IcpArgs& a, int iter, int lane, double* sums, double* tmp)
{
    const unsigned long long* __restrict__ buf = a.acc + (size_t)(iter % 3) * (ICP_SHARDS * 56);
    unsigned long long v[ICP_SHARDS];
#pragma unroll
    for (int h = 0; h < ICP_SHARDS; h++) v[h] = buf[h * 56 + min(lane, 55)];
    unsigned long long tot = 0ull;
#pragma unroll
    for (int h = 0; h < ICP_SHARDS; h++) tot += v[h];
    if (lane < 56) tmp[lane] = ldexp((double)(long long)tot, lane < 28 ? -16 : -60);
    __threadfence_block(); __builtin_amdgcn_wave_barrier();      // one wavefront: its LDS operations complete in order
    if (lane < 28) sums[lane] = tmp[lane] + tmp[28 + lane];
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
}

constexpr int ICP_QPB = ICP_BLOCK / KNN_G;        // features per workgroup of lvi_debug_residuals' kernel

// round-to-nearest-even of |x| < 2^51 to an integer with one f64 add (as in lvi_voxel.hip)
__device__ __forceinline__ long long d2ll_rn_small_icp(double x)
{
    const double M = 6755399441055744.0;                            // 2^52 + 2^51
    return __double_as_longlong(x + M) - __double_as_longlong(M);
}
// exact fixed point of one partial sum: v = coarse 2^-16 + fine 2^-60, |v| < 2^45 (beyond: saturated)
__device__ __forceinline__ void fx_split(double v, long long& coarse, long long& fine)
{
    v = fmin(fmax(v, -3.5184372088832e13), 3.5184372088832e13);                 // +-2^45
    const double c = rint(ldexp(v, 16));
    coarse = (long long)c;
    fine = d2ll_rn_small_icp(ldexp(v - ldexp(c, -16), 60));                      // |v - c 2^-16| <= 2^-17: the product is below 2^44
}

// One launch = one Gauss-Newton iteration of every scan of the batch (blockIdx.z): the end of the PREVIOUS iteration (solve,
// pose update, convergence test — from the exact totals the previous launch left), then residuals, rows and normal equations
// of this one.  A workgroup owns QPB features (64 in iteration 0, where every feature searches the unit ball and the chip wants
// many workgroups; 256 afterwards, where a workgroup is four wavefronts of one lane per feature and the head of the launch is
// paid once per 256 features):
//   H  first wavefront: totals of the previous launch -> 6 x 6 solve -> pose of this iteration (every workgroup: no exchange)
//   0  one lane per feature: the feature's record (its five neighbours with their coordinates, where it stood at its last
//      search, its fit) arrives in ONE round of coalesced loads, requested before H starts; the skip test of the 5-NN search; the
//      features that search are listed
//   A  G lanes per LISTED feature, NT / G features per round: 5-NN; a result that differs from the record fetches the new
//      neighbours' coordinates
//   B  one lane per feature: line / plane fit when the five or their order changed (else the stored fit), evaluation at the
//      current pose, the Gauss-Newton row (6 + 1 floats) into LDS
//   C  the 27 products of the rows are formed in f64 and added up in a fixed shape; the 28 sums join the launch's exact
//      fixed-point accumulators (integer adds commute: the totals do not depend on arrival order, on the lanes per feature or
//      on the XCD placement).  Nobody waits for the adds: the next launch reads the totals.
template <int QPB, bool FIRST, int G, int KB, bool TILES = false>
__global__ __launch_bounds__(QPB == 64 ? 64 * G : 256, TILES ? 1 : 4) void icp_gn_kernel(Batch<IcpArgs> B_, int iter)
{
    constexpr int NT = QPB == 64 ? 64 * G : 256;
    static_assert(QPB == 64 || QPB == 256, "features per workgroup");
    static_assert(!FIRST || QPB == 64, "iteration 0 (no previous iteration to end, no records to read) runs the 64-feature form");
    static_assert(NT >= 128 && NT >= QPB, "phase C uses 128 threads; phases 0 and B one lane per feature");
    const IcpArgs& a = B_.a[blockIdx.z];
    IcpState& st = *a.st;
    // everything the head needs is requested before the first value is tested (one round trip instead of four in a row)
    const int done = st.done;
    const int nC = a.nq[0], nS = a.nq[1];
    const IcpPose& pin = st.pose[(iter + 1) & 1];   // iteration 0: buffer 0 holds the initial guess (set_pose_init); iteration i: the pose launch i - 1 worked with
    const IcpPose& p0 = st.pose[0];
    const float poseA = threadIdx.x < 12 ? p0.A[threadIdx.x] : 0.f, poseT = threadIdx.x < 6 ? p0.trig[threadIdx.x] : 0.f;
    const float tin = pin.T[threadIdx.x % 6];
    const int degen_in = st.degenerate;
    if (done) return;                                // the loop is over (icp_final_kernel finishes the scan)
    if (FIRST && !(a.have_map && nC > a.edgeMin && nS > a.surfMin && nC + nS > 0)) {   // :1317, :1320 (the same in every workgroup)
        if (blockIdx.x == 0 && threadIdx.x == 0) { st.done = 1; st.status = a.have_map ? LVI_TOO_FEW_FEATURES : LVI_NO_MAP; }
        return;
    }
    // XCD-aware placement (speed only): workgroups b and b + 8 share an XCD and its L2, so XCD k takes a CONTIGUOUS eighth of the
    // features — they come in voxel order, a spatial slab — and its L2 fetches that slab's part of the index and the map instead
    // of all of it
    const int Q = nC + nS;
    const int nb = (Q + QPB - 1) / QPB;
    const int per_xcd = (nb + 7) / 8;
    // The grid need not cover the features: a workgroup goes on to the block `stride` further (inside its XCD's share) until the scan's
    // blocks are used up.  The host sizes the grid from the feature counts the last finished matches reported (gn_grid_features), not
    // from the capacity: eight scans' worth of capacity were ~7 000 workgroups per launch that allocated registers and LDS to find out
    // they had nothing to do (+2 % scans/s under four-handle load).  Any grid gives the same sums (exact integers, shard = block & 7).
    int jx = a.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int jstride = a.xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int jlim = a.xcd_map ? per_xcd : nb;
    const int jbase = a.xcd_map ? (int)(blockIdx.x & 7u) * per_xcd : 0;
    if (jx >= jlim || jbase + jx >= nb) return;
    __shared__ float sA[12], sT[6];
    __shared__ __attribute__((aligned(16))) float srow[QPB][8];                  // the Gauss-Newton row of a feature: matA(i, 0..5), matB(i), selected (1 / 0)
    __shared__ long long spi[QPB / 16][56];         // exact fixed-point images (coarse, fine) of the sums of 16 consecutive features
    __shared__ double ssum[28], stmp[56];
    __shared__ float sd4[QPB];                      // squared distance of the fifth nearest (the callers' gate)
    __shared__ int si[5][QPB];                      // their map indices
    __shared__ float sori[4][QPB];
    __shared__ unsigned short sperm[QPB];           // position in the record of the j-th nearest (3 bits each)
    __shared__ unsigned char sfok[QPB], srefit[QPB];
    __shared__ float sr2[QPB];                      // search radius (squared) of the features that search, in list order
    __shared__ unsigned short slist[QPB];           // the features (workgroup-local) that search in this iteration
    __shared__ int swcnt[QPB / 64];                 // searching features per wavefront
    constexpr int TP = FIRST ? 384 : 256, TC = FIRST ? 1024 : 768;      // points / cell starts of a wavefront's index tile (unit ball | bounded ball)
    __shared__ __attribute__((aligned(16))) KnnTile<TP, TC> stile[TILES ? NT / 64 : 1];      // (the tile form is a separate instantiation: LVI_KNN_TILES=1)
    __shared__ int sdone;
    // (LDS is declared out here: a __shared__ variable inside the body below would exist once per instantiation)
    // (the block body is instantiated twice — with the head of the launch for a workgroup's first block, without it for the blocks it
    // walks on to: as ONE loop body the head's registers stayed live around the loop and the kernel spilled 250 bytes per lane)
    auto run_block = [&](const int wg, auto first_tag) -> bool {
    constexpr bool first = decltype(first_tag)::value;
    const bool stamp = (wg == nb / 2 && threadIdx.x == 0 && (a.stamp_iter < 0 || a.stamp_iter == iter));      // a surf workgroup in the middle
    long long t_prev = stamp ? clock64() : 0, t_first = t_prev, cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define LVI_STAMP(slot) do { if (stamp) { const long long t_now = clock64(); cyc[slot] += t_now - t_prev; t_prev = t_now; } } while (0)
    const bool use_prev = !FIRST && a.nn_prev != nullptr;
    const int cap = a.cap;
    constexpr unsigned PERM_ID = 0u | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12);
    // ---- phase 0, first half: the loads of the feature's record, in flight while the head of the launch runs
    const int ql = threadIdx.x;
    const int t = wg * QPB + ql;
    const bool lane_feat = ql < QPB;
    const bool active = lane_feat && t < Q;
    const int tc = (lane_feat && t < Q) ? t : Q - 1;
    lvi_pt ori = {0.f, 0.f, 0.f, 0.f};
    int id[5] = {-1, -1, -1, -1, -1};
    float4 pp[5];
    float4 ref = make_float4(0.f, 0.f, 0.f, 0.f), f1 = ref, f2 = ref;
    unsigned fok = 0u;
#pragma unroll
    for (int j = 0; j < 5; j++) pp[j] = ref;
    if (lane_feat) {
        const lvi_pt* qp = tc < nC ? a.q[0] + tc : a.q[1] + (tc - nC);
        ori = *qp;
        if (use_prev) {                                     // uniform: one round of loads, all in flight together
#pragma unroll
            for (int j = 0; j < 5; j++) { id[j] = a.nn_prev[(size_t)j * cap + tc]; pp[j] = a.nn_pt[(size_t)j * cap + tc]; }
            if (a.nn_ref) ref = a.nn_ref[tc];
            f1 = a.fit[tc]; f2 = a.fit2[tc]; fok = a.fit_ok[tc];
        }
    }
    // ---- H: the end of iteration iter - 1 (once per workgroup: its first block)
    if constexpr (first) {
    if constexpr (FIRST) {
        if (threadIdx.x < 12) sA[threadIdx.x] = poseA;
        if (threadIdx.x < 6) sT[threadIdx.x] = poseT;
        if (threadIdx.x == 0) sdone = 0;
    } else if (threadIdx.x < 64) {
        // The totals the previous launch left (exact integers: the same in every workgroup, whatever order they were added in),
        // the 6 x 6 solve, the new pose.  No workgroup waits for another one.
        const long long tq0 = stamp ? clock64() : 0;
        IcpPose P;
#pragma unroll
        for (int k = 0; k < 6; k++) P.T[k] = rdl(tin, k);
        icp_take_sums(a, iter - 1, threadIdx.x, ssum, stmp);
        const long long tq1 = stamp ? clock64() : 0;
        const int dn = icp_iter_end(a, iter - 1, ssum, threadIdx.x, wg == 0, degen_in, P);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) sA[k] = P.A[k];
#pragma unroll
            for (int k = 0; k < 6; k++) sT[k] = P.trig[k];
            sdone = dn;
        }
        if (stamp) { const long long tq2 = clock64(); a.cyc[8] = tq1 - tq0; a.cyc[9] = tq2 - tq1; a.cyc[12] = tq2 - tq0; }
    }
    // the buffer the NEXT launch adds into (last read by the previous launch)
    if (wg == 0) { unsigned long long* z = a.acc + (size_t)((iter + 1) % 3) * (ICP_SHARDS * 56); for (int k = threadIdx.x; k < ICP_SHARDS * 56; k += NT) z[k] = 0ull; }
    __syncthreads();
    if (sdone) return true;
    }
    LVI_STAMP(0);
    // ---- phase 0, second half.  From the second iteration on a feature knows its previous five neighbours; their distances
    // under the new pose bound the fifth-nearest distance (five map points lie inside that ball), and the search either shrinks
    // to that ball or is not needed at all: the feature's last search left a lower bound LB on the distance from where it stood
    // THEN (ref) to every map point outside its five, so every such point is at least LB - |sel - ref| away now; if that exceeds
    // the farthest of the five (with 2e-4 m of room for the f32 rounding of the distances, 1e-6 relative), the five are still
    // the five nearest, no outsider can even tie, and their (distance, index) order is recomputed here with the search's own
    // expression: the same Knn5, bit for bit, without a search.
    if (lane_feat) {
        const lvi_pt sel = to_map(sA, ori);
        bool need = active;
        float r2 = KNN_R2_FULL;
        unsigned perm = PERM_ID;
        KnnKeys kk;
#pragma unroll
        for (int j = 0; j < 5; j++) kk.k[j] = KNN_EMPTY;
        if (active && use_prev && id[0] >= 0) {
            float b = 0.f;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const float ex = sub_rn(sel.x, pp[j].x), ey = sub_rn(sel.y, pp[j].y), ez = sub_rn(sel.z, pp[j].z);
                const float dist = add_rn(add_rn(mul_rn(ex, ex), mul_rn(ey, ey)), mul_rn(ez, ez));
                b = fmaxf(b, dist);
                // (distance, index) order; the record position rides in the low three bits (indices are below 2^25)
                knn_insert(kk, ((unsigned long long)__float_as_uint(dist) << 32) | ((unsigned)id[j] << 3) | (unsigned)j);
            }
            if (b < 1.0f) {
                const float rb = sqrtf(b);
                if (a.nn_ref) {
                    const float dx = sel.x - ref.x, dy = sel.y - ref.y, dz = sel.z - ref.z;
                    if (rb + sqrtf(dx * dx + dy * dy + dz * dz) + 2e-4f < sqrtf(ref.w)) need = false;
                }
                const float rs = rb + a.knn_slack;
                r2 = fminf(fmaxf(b, rs * rs), KNN_R2_FULL);
            }
        }
        if (!need && active) {
            perm = 0u;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const unsigned lo = (unsigned)kk.k[j];
                if (j == 4) sd4[ql] = __uint_as_float((unsigned)(kk.k[j] >> 32));
                si[j][ql] = (int)(lo >> 3);
                perm |= (lo & 7u) << (3 * j);
            }
            srefit[ql] = (perm != PERM_ID || fok == 0u) ? 1 : 0;
        }
        if (!active) {
#pragma unroll
            for (int j = 0; j < 5; j++) si[j][ql] = -1;
            sd4[ql] = INFINITY;
            srefit[ql] = 0;
        }
        sori[0][ql] = ori.x; sori[1][ql] = ori.y; sori[2][ql] = ori.z; sori[3][ql] = ori.intensity;
        sperm[ql] = (unsigned short)perm; sfok[ql] = (unsigned char)fok;
        // the searching features, listed in feature order: rank inside the wavefront now, the wavefronts' offsets after the barrier
        const unsigned long long mk = __ballot(need);
        const int wv = ql >> 6;
        if ((ql & 63) == 0) swcnt[wv] = __popcll(mk);
        if (QPB == 64) { if (need) { const int pos = __popcll(mk & ((1ull << (ql & 63)) - 1ull)); slist[pos] = (unsigned short)ql; sr2[pos] = r2; } }
        else {
            __syncthreads();                        // (QPB == NT here: every thread is a feature lane and reaches this barrier)
            int off = 0;
#pragma unroll
            for (int w = 0; w < QPB / 64; w++) off += w < wv ? swcnt[w] : 0;
            if (need) { const int pos = off + __popcll(mk & ((1ull << (ql & 63)) - 1ull)); slist[pos] = (unsigned short)ql; sr2[pos] = r2; }
        }
    }
    __syncthreads();
    int nsearch = 0;
#pragma unroll
    for (int w = 0; w < QPB / 64; w++) nsearch += swcnt[w];
    LVI_STAMP(6);
    // ---- phase A: G lanes per SEARCHING feature, NT / G features per round
    {
        long long tk[6] = {0, 0, 0, 0, 0, 0};
        float r2 = KNN_R2_FULL;
        for (int base = 0; base < nsearch; base += NT / G) {
            const int gi = base + threadIdx.x / G, sub = threadIdx.x % G;
            const bool act = gi < nsearch;
            const int qs = act ? slist[gi] : 0;
            const int ts = wg * QPB + qs;
            const int w = ts < nC ? 0 : 1;
            lvi_pt o2; o2.x = sori[0][qs]; o2.y = sori[1][qs]; o2.z = sori[2][qs]; o2.intensity = 0.f;
            const lvi_pt sel = to_map(sA, o2);
            r2 = act ? sr2[gi] : 0.f;
            Knn5 r;
            float lb2 = 0.f;
            // the wavefront's features share an LDS tile of the index when they are of one kind (corner | surf) and the tile fits
            const unsigned long long mc = __ballot(act && w == 0), ms = __ballot(act && w == 1);
            bool tiled = false;
            KnnTileAcc<TP, TC> tacc;
            const int wu = mc ? 0 : 1;
            if (TILES && (mc == 0ull) != (ms == 0ull))
                tiled = knn_tile_build<TP, TC>(*a.meta[wu], a.cell_start[wu], a.sorted[wu], act, sel.x, sel.y, sel.z, r2, stile[threadIdx.x >> 6], tacc);
            if (TILES && tiled) knn5_search_acc<G, KB>(*a.meta[wu], tacc, act, sel.x, sel.y, sel.z, sub, r, stamp ? tk : nullptr, r2, a.nn_ref != nullptr, &lb2);
            else if (act) knn5_search_group<G, KB>(*a.meta[w], a.cell_start[w], a.sorted[w], sel.x, sel.y, sel.z, sub, r, stamp ? tk : nullptr, r2, a.nn_ref != nullptr, &lb2);
            if (TILES && (threadIdx.x & 63) == 0 && (mc | ms)) atomicAdd((unsigned long long*)&a.cyc[tiled ? 11 : 10], 1ull);   // wavefront rounds on a tile / on the direct path
            if (act && sub == 0) {
                const bool five = r.d[4] < 1.0f;
                // the same five in the same order as the record (a search usually confirms them): its coordinates and its fit stand
                bool same = five && use_prev && sfok[qs] != 0;
                if (same) {                                    // (the record's indices: phase 0 of this workgroup just read them, an L1 hit)
                    int idr[5];
#pragma unroll
                    for (int j = 0; j < 5; j++) idr[j] = a.nn_prev[(size_t)j * cap + ts];
#pragma unroll
                    for (int j = 0; j < 5; j++) same = same && r.i[j] == idr[j];
                }
                if (!same && a.nn_prev) {
                    // the new neighbours become the feature's record right here (phase B reads their coordinates back after the barrier)
                    const lvi_pt* __restrict__ map = a.mapds[w];
                    lvi_pt nbp[5];
#pragma unroll
                    for (int j = 0; j < 5; j++) nbp[j] = map[five ? r.i[j] : 0];           // unconditional: the five loads travel together
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        a.nn_prev[(size_t)j * cap + ts] = five ? r.i[j] : -1;
                        a.nn_pt[(size_t)j * cap + ts] = make_float4(nbp[j].x, nbp[j].y, nbp[j].z, 0.f);
                    }
                }
#pragma unroll
                for (int j = 0; j < 5; j++) si[j][qs] = r.i[j];
                sd4[qs] = r.d[4];
                sperm[qs] = (unsigned short)PERM_ID;
                srefit[qs] = same ? 0 : 1;
                if (a.nn_ref) a.nn_ref[ts] = make_float4(sel.x, sel.y, sel.z, five ? lb2 : 0.f);
            }
        }
        LVI_STAMP(1);
        if (stamp) { cyc[7] = tk[2] - tk[1]; cyc[2] = tk[3] - tk[2]; a.cyc[13] = tk[5]; a.cyc[14] = r2 < KNN_R2_FULL ? 1 : 0; }
        if (threadIdx.x == 0) atomicAdd((unsigned long long*)&a.cyc[15], (unsigned long long)nsearch);      // searches of this scan match, all iterations
    }
    __syncthreads();
    // ---- phase B, one lane per feature
    if (lane_feat) {
        bool ok = false;
        float rA[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rB = 0.f;
        if (t < Q) {
            const bool isC = t < nC;
            const lvi_pt sel = to_map(sA, ori);
            const bool five = sd4[ql] < 1.0f;                                    // :1025, :1121
            const bool refit = srefit[ql] != 0;
            const unsigned perm = sperm[ql];
            fok = sfok[ql];
            if (refit) {
                // the five, in (distance, index) order: from the feature's record (a searching feature's was rewritten by phase A;
                // a feature whose order changed under the new pose reads its record through the permutation), or — without
                // records (LVI_KNN_NO_BOUND) — from the map
                lvi_pt nb[5];
                const lvi_pt* __restrict__ map = a.mapds[isC ? 0 : 1];
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int m = (perm >> (3 * j)) & 7u;
                    if (a.nn_prev) { const float4 v = a.nn_pt[(size_t)m * cap + t]; nb[j].x = v.x; nb[j].y = v.y; nb[j].z = v.z; }
                    else { const lvi_pt v = map[five ? si[j][ql] : 0]; nb[j].x = v.x; nb[j].y = v.y; nb[j].z = v.z; }
                    nb[j].intensity = 0.f;
                }
                if (five) {
                    bool valid;
                    if (isC) valid = corner_fit(nb, f1, f2); else valid = surf_fit(nb, f1);
                    fok = valid ? 1u : 2u;
                } else fok = 0u;
                if (a.nn_prev) {
                    if (perm != PERM_ID) {                  // the record follows the fit: neighbours in the fit's order, with their coordinates
#pragma unroll
                        for (int j = 0; j < 5; j++) {
                            a.nn_prev[(size_t)j * cap + t] = si[j][ql];
                            a.nn_pt[(size_t)j * cap + t] = make_float4(nb[j].x, nb[j].y, nb[j].z, 0.f);
                        }
                    }
                    a.fit[t] = f1; a.fit_ok[t] = (unsigned char)fok;
                    if (isC) a.fit2[t] = f2;
                }
            }
            lvi_pt cf = {0.f, 0.f, 0.f, 0.f};
            if (five && fok == 1u) ok = isC ? corner_eval(f1, f2, sel, cf) : surf_eval(f1, ori, sel, cf);
            if (ok) lm_row(sT, ori, cf, rA, rB);
        }
        float4* rw = reinterpret_cast<float4*>(&srow[ql][0]);
        rw[0] = ok ? make_float4(rA[0], rA[1], rA[2], rA[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        rw[1] = ok ? make_float4(rA[4], rA[5], rB, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    LVI_STAMP(3);
    __syncthreads();
    // ---- phase C: column k of the 28 (21 products of AtA, 6 of AtB, the count) over 16 consecutive features, in feature order,
    // in f64; that sum is turned into exact fixed point, and from there on everything is integer: the totals of a launch do not
    // depend on the features per workgroup, on the lanes per feature, on the XCD placement or on arrival order
    constexpr int NG = QPB / 16;                    // groups of 16 features (aligned to the feature index)
    for (int idx = threadIdx.x; idx < NG * 32; idx += NT) {
        const int k = idx & 31, g = idx >> 5;
        if (k < 28) {
            int r = (k >= 6) + (k >= 11) + (k >= 15) + (k >= 18) + (k >= 20);
            int c = r + (k - (r * 6 - (r * (r - 1)) / 2));
            if (k >= 21) { r = k - 21; c = 6; }
            if (k == 27) { r = 7; c = 7; }
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 16; q++) v += (double)srow[g * 16 + q][r] * (double)srow[g * 16 + q][c];
            long long co, fi;
            fx_split(v, co, fi);
            spi[g][k] = co; spi[g][28 + k] = fi;
        }
    }
    __syncthreads();
    if (threadIdx.x < 56) {
        long long tot = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) tot += spi[g][threadIdx.x];
        unsigned long long* dst = a.acc + (size_t)(iter % 3) * (ICP_SHARDS * 56) + (size_t)(wg & (ICP_SHARDS - 1)) * 56;
        if (tot) atomicAdd(&dst[threadIdx.x], (unsigned long long)tot);       // no one waits for these: the next launch reads the totals
    }
    LVI_STAMP(4);
    if (stamp) { cyc[5] = clock64() - t_first; for (int q = 0; q < 8; q++) a.cyc[q] = cyc[q]; }
    return false;
    };
#undef LVI_STAMP
    if (run_block(jbase + jx, std::true_type{})) return;
    for (jx += jstride; jx < jlim && jbase + jx < nb; jx += jstride) {
        __syncthreads();                            // the previous block's LDS (rows, lists, sums) is done with
        (void)run_block(jbase + jx, std::false_type{});
    }
}

// After the last Gauss-Newton launch: the end of its iteration (unless the loop ended earlier), transformUpdate, the pose
// record.  One wavefront per scan.
__global__ __launch_bounds__(64) void icp_final_kernel(Batch<IcpArgs> B_, int n_iters)
{
    const IcpArgs& a = B_.a[blockIdx.z];
    IcpState& s = *a.st;
    __shared__ double ssum[28], tmp[56];
    if (n_iters <= 0) {                                                            // no loop at all: the gates alone decide the status
        if (threadIdx.x == 0) {
            if (!a.have_map) s.status = LVI_NO_MAP;                                                           // :1317
            else if (!(a.nq[0] > a.edgeMin && a.nq[1] > a.surfMin)) s.status = LVI_TOO_FEW_FEATURES;          // :1320
        }
    } else if (!s.done) {
        IcpPose P;
        const IcpPose& pin = s.pose[(n_iters - 1) & 1];
#pragma unroll
        for (int k = 0; k < 6; k++) P.T[k] = pin.T[k];
        icp_take_sums(a, n_iters - 1, threadIdx.x, ssum, tmp);
        (void)icp_iter_end(a, n_iters - 1, ssum, threadIdx.x, true, s.degenerate, P);
    }
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) { icp_finish_body(a); if (a.h_feat) *a.h_feat = a.nq[0] + a.nq[1]; }
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    if (a.h_res) {
        // the result block goes out a word per lane and round (81 posted writes over PCIe from ONE lane were 7 us of this launch)
        static_assert(sizeof(IcpHostResult) % 4 == 0, "words");
        constexpr int NW = (int)(sizeof(IcpHostResult) / 4), M = LVI_ICP_MAX_ITERS;
        for (int w = threadIdx.x; w < NW; w += 64) {
            int v = 0;
            if (w == 0) v = s.final_status; else if (w == 1) v = s.iters; else if (w == 2) v = s.converged; else if (w == 3) v = s.degenerate;
            else if (w == 4) v = s.done; else if (w == 5) v = s.status;
            else if (w < 6 + M) v = s.n_sel[w - 6];
            else if (w < 12 + M) v = __float_as_int(s.final_pose[w - 6 - M]);
            else if (w < 15 + M) v = a.nout3[w - 12 - M];
            else if (w < 17 + M) v = a.d_status[w - 15 - M];
            reinterpret_cast<int*>(a.h_res)[w] = v;
        }
    }
}

__global__ __launch_bounds__(256) void transform_kernel(const lvi_pt* in, int n, IcpPose pose, lvi_pt* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = to_map(pose.A, in[i]);
}

__global__ void pose_only_kernel(IcpPose* out, float t0, float t1, float t2, float t3, float t4, float t5)
{
    IcpPose p;
    p.T[0] = t0; p.T[1] = t1; p.T[2] = t2; p.T[3] = t3; p.T[4] = t4; p.T[5] = t5;
    make_pose(p);
    *out = p;
}

__global__ __launch_bounds__(256) void transform_dev_pose_kernel(const lvi_pt* in, int n, const IcpPose* pose, lvi_pt* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = to_map(pose->A, in[i]);
}

__global__ __launch_bounds__(ICP_BLOCK) void residual_debug_kernel(IcpArgs a, int which, const IcpPose* pose, lvi_pt* coeff, uint8_t* flag)
{
    const int n = a.nq[which];
    const int ql = threadIdx.x / KNN_G, sub = threadIdx.x % KNN_G;
    const int t = blockIdx.x * ICP_QPB + ql;
    if (blockIdx.x * ICP_QPB >= n) return;
    const bool active = t < n;
    lvi_pt ori = {0.f, 0.f, 0.f, 0.f};
    if (active) ori = a.q[which][t];
    const lvi_pt sel = to_map(pose->A, ori);
    Knn5 r;
    if (active) knn5_search_group<KNN_G, 8>(*a.meta[which], a.cell_start[which], a.sorted[which], sel.x, sel.y, sel.z, sub, r);
    if (active && sub == 0) {
        lvi_pt cf = {0.f, 0.f, 0.f, 0.f};
        const bool ok = which == 0 ? corner_residual(a, sel, r, cf) : surf_residual(a, ori, sel, r, cf);
        flag[t] = ok ? 1 : 0;
        coeff[t] = ok ? cf : lvi_pt{0.f, 0.f, 0.f, 0.f};
    }
}

__global__ void set_dyn2_kernel(VoxSegDyn* dyn, int n0, int n1)
{
    dyn[0].in_off = 0; dyn[0].n = n0;
    dyn[1].in_off = 0; dyn[1].n = n1;
}

IcpArgs icp_args(LidarDev& d)
{
    IcpArgs a{};
    a.st = d.icp;
    a.q[0] = d.cornerDS; a.q[1] = d.surfDS; a.nq = d.voxScan.d_nout;
    for (int w = 0; w < 2; w++) { a.meta[w] = d.grid[w].meta; a.cell_start[w] = d.grid[w].cell_start; a.sorted[w] = d.grid[w].sorted; }
    a.mapds[0] = d.mapCornerDS; a.mapds[1] = d.mapSurfDS;
    a.cyc = d.d_icp_cycles; a.d_status = d.d_status; a.nn_prev = d.knn_bound ? d.nnPrev : nullptr;
    a.cap = d.ext_cap; a.nn_pt = d.nnPt; a.fit = d.fitA; a.fit2 = d.fitB; a.fit_ok = d.fitOk; a.acc = d.icpAcc;
    a.nn_ref = (d.knn_bound && d.knn_skip) ? d.nnRef : nullptr; a.knn_slack = d.knn_slack;
    { static const bool no_map = getenv("LVI_ICP_NO_XCD_MAP") != nullptr; a.xcd_map = no_map ? 0 : 1; }
    a.stamp_iter = d.icp_stamp_iter; a.lds_tiles = d.knn_tiles ? 1 : 0; a.h_feat = d.h_gn_feat; a.h_res = d.h_res; a.nout3 = d.voxScan.d_nout;
    a.edgeMin = d.P.edgeFeatureMinValidNum; a.surfMin = d.P.surfFeatureMinValidNum;
    a.max_iters = std::min(d.P.icp_max_iters, LVI_ICP_MAX_ITERS); a.disable_break = d.P.icp_disable_break;
    a.rot_tol = d.P.rotation_tollerance; a.z_tol = d.P.z_tollerance; a.imu_weight = (double)d.P.imuRPYWeight;
    return a;
}

}  // namespace

// everything enqueued on the main stream so far must finish before a later local-map update (second stream) touches the map
// buffers or reads the keyframe store
void mark_map_deps(LidarDev& d)
{
    if (!d.evMapDeps) LVI_HIP(hipEventCreateWithFlags(&d.evMapDeps, hipEventDisableTiming));
    LVI_HIP(hipEventRecord(d.evMapDeps, d.ctx.stream));
    d.have_map_deps = true;
}

void join_map(LidarDev& d)
{
    if (d.map_pending) { LVI_HIP(hipStreamWaitEvent(d.ctx.stream, d.evMap, 0)); d.map_pending = false; }
}

static GridArgs grid_args(LidarDev& d)
{
    GridArgs g{};
    for (int w = 0; w < 2; w++) {
        g.meta[w] = d.grid[w].meta; g.cell_start[w] = d.grid[w].cell_start; g.sorted[w] = d.grid[w].sorted;
        g.count[w] = d.grid[w].count; g.blockSum[w] = d.grid[w].blockSum;
    }
    g.vox = d.voxMap.d_grid; g.nout = d.voxMap.d_nout;
    g.ds[0] = d.mapCornerDS; g.ds[1] = d.mapSurfDS;
    g.cap = d.map_cap; g.max_cells = d.max_cells; g.d_status = d.d_status;
    return g;
}

// The map build does not depend on the current scan, so it runs on its own stream and overlaps the
// scan-side stages (organise / sector kernel / scan voxel grids, which occupy only a few CUs); the
// main stream joins it right before scan matching.  Batch: every slot re-voxelises and re-indexes the (shared) raw map
// into its own DS map and index, as the reference does for every scan.
void stage_map_build(const Slots& sl)
{
    LidarDev& d = sl.first();
    const Ctx& cx = d.P.map_on_main_stream ? d.ctx : d.ctx2;
    // everything already enqueued on the main stream (map upload, the previous scan's GN loop reading the
    // previous index) must finish before the map buffers are rewritten
    const bool forked = cx.stream != d.ctx.stream;
    if (forked) {
        LVI_HIP(hipEventRecord(d.evMain, d.ctx.stream));
        LVI_HIP(hipStreamWaitEvent(cx.stream, d.evMain, 0));
    }
    const bool cache_plan = d.P.map_plan_cache != 0;
    // map_plan_cache = 1: bbox of the raw map, per-bin counts and partition offsets are taken once, when the map was (re)written (PCL's
    // getMinMax3D is a pure function of the unchanged input).  Default (0): every re-voxelisation takes them again, per slot, as
    // the reference's VoxelGrid::filter does for every scan (mapOptimization.cpp:958-965) — in one pass (vb_plan).
    if (cache_plan && (!d.voxMap.bbox_cached || (!d.voxMap.hist_cached && voxel_resolve_mode(d.voxMap) == VOX_BINNED))) {      // (AUTO: the first build of a plan is sorted)
        d.voxMap.n_host[0] = d.n_map_corner; d.voxMap.n_host[1] = d.n_map_surf; d.voxMap.use_n_host = true;
        voxel_bbox_pass(cx, d.voxMap, "map", (double)d.n_map_corner + (double)d.n_map_surf);
        d.voxMap.bbox_cached = true;
    }
    // raw map counts are host-known here; the voxel plan wants them in device memory
    const VoxelPlan* plans[MAX_BATCH];
    for (int z = 0; z < sl.n; z++) {
        LidarDev& q = sl[z];
        q.n_map_corner = d.n_map_corner; q.n_map_surf = d.n_map_surf;
        q.voxMap.n_host[0] = d.n_map_corner; q.voxMap.n_host[1] = d.n_map_surf; q.voxMap.use_n_host = true;       // instead of a 1-thread launch writing d_dyn
        q.voxMap.plan_per_run = !cache_plan;
        q.voxMap.bbox_cached = cache_plan; q.voxMap.hist_cached = cache_plan && d.voxMap.hist_cached;
        plans[z] = &q.voxMap;
        q.have_map = true;
    }
    const double n = ((double)d.n_map_corner + (double)d.n_map_surf) * sl.n;
    voxel_downsample_batch(cx, plans, sl.n, "map", n);
    stage_map_index(sl, cx);
    if (forked) {
        LVI_HIP(hipEventRecord(d.evMap, cx.stream));
        d.map_pending = true;
    }
}
void stage_map_build(LidarDev& d) { stage_map_build(OneSlot(d).s); }

// the replacement of the two KdTreeFLANN::setInputCloud calls (mapOptimization.cpp:1322-1323) over every slot's DS map
void stage_map_index(const Slots& sl, const Ctx& cx)
{
    Batch<GridArgs> G;
    double nds = 0;
    for (int z = 0; z < sl.n; z++) { G.a[z] = grid_args(sl[z]); nds += 0.02 * ((double)sl[z].n_map_corner + (double)sl[z].n_map_surf); }   // nominal DS size, byte accounting only
    for (int z = sl.n; z < MAX_BATCH; z++) G.a[z] = G.a[0];
    const unsigned S = (unsigned)sl.n;
    LVI_LAUNCH(cx, "grid_count", 16.0 * nds, hipLaunchKernelGGL(grid_count_kernel, dim3(GRID_PT_BLOCKS, 2, S), dim3(256), 0, cx.stream, G));
    LVI_LAUNCH(cx, "grid_scan_sum", 0, hipLaunchKernelGGL(grid_scan_sum_kernel, dim3(GRID_SCAN_BLOCKS, 2, S), dim3(256), 0, cx.stream, G));
    LVI_LAUNCH(cx, "grid_scan_apply", 0, hipLaunchKernelGGL(grid_scan_apply_kernel, dim3(GRID_SCAN_BLOCKS, 2, S), dim3(256), 0, cx.stream, G));
    LVI_LAUNCH(cx, "grid_scatter", 32.0 * nds, hipLaunchKernelGGL(grid_scatter_kernel, dim3(GRID_PT_BLOCKS, 2, S), dim3(256), 0, cx.stream, G));
}

static void kf_matrix(const float* T, float M[12])
{
    // pcl::getTransformation(x, y, z, roll, pitch, yaw), transformIn = [roll, pitch, yaw, x, y, z] (:404-407), host libm as the reference
    const float A = std::cos(T[2]), B = std::sin(T[2]), C = std::cos(T[1]), D = std::sin(T[1]), E = std::cos(T[0]), F = std::sin(T[0]), DE = D * E, DF = D * F;
    const float R[12] = {A * C, A * DF - B * E, B * F + A * DE, T[3],  B * C, A * E + B * DF, B * DE - A * F, T[4],  -D, C * F, C * E, T[5]};
    for (int q = 0; q < 12; q++) M[q] = R[q];
}

// f-4, incremental form.  The key list of this scan against the keyframes the tables already hold: whole keyframes enter
// (+1) or leave (-1), with multiplicity (the reference's list may name a key twice, :921-927); then the live voxels are
// emitted for the current bounding box and indexed.  Everything is enqueued on the main stream; one 16-byte read tells
// the host whether the device accepted (range, table size, PCL's overflow rule) — a sequential node reads the pose of
// every scan anyway.
bool stage_map_update(LidarDev& d, const int32_t* keys, int n_keys)
{
    IncMap& m = d.inc;
    if (!m.H || n_keys > m.max_active) return false;
    join_map(d);
    // like the full build, the update runs on the handle's second stream unless map_on_main_stream: it overlaps the scan-side
    // stages already enqueued on the main stream, and the host waits for THIS stream only
    const Ctx& cx = d.P.map_on_main_stream ? d.ctx : d.ctx2;
    const bool forked = cx.stream != d.ctx.stream;
    if (forked) {
        // What the update must wait for on the main stream: the previous scan's GN loop (it reads the index this update rewrites)
        // and the keyframe copies into the store — recorded where they were enqueued (mark_map_deps).  NOT the stages of the
        // CURRENT scan already in that stream (organise, sector kernel, the scan's grids): the update runs beside them.
        if (d.have_map_deps) LVI_HIP(hipStreamWaitEvent(cx.stream, d.evMapDeps, 0));
    }
    const int nkf = (int)d.kf_pose.size();
    std::vector<int> want(nkf, 0);
    for (int i = 0; i < n_keys; i++) want[keys[i]]++;
    d.inc_mult.resize(nkf, 0); d.inc_pose.resize(nkf);
    // a pose corrected since the key was added (correctPoses :1650-1660), or too many dead voxels: start over
    bool rebuild = !d.inc_ready || d.inc_nocc_bound > m.H / 2;
    for (int k = 0; k < nkf && !rebuild; k++)
        if (d.inc_mult[k] > 0 && want[k] > 0 && std::memcmp(d.inc_pose[k].data(), d.kf_pose[k].data(), sizeof(float) * 6) != 0) rebuild = true;
    if (rebuild) { incmap_clear(cx, m); std::fill(d.inc_mult.begin(), d.inc_mult.end(), 0); d.inc_nocc_bound = 0; }
    int np = 0, maxn = 1;
    long long added = 0;
    for (int k = 0; k < nkf; k++) {
        const int delta = want[k] - d.inc_mult[k];
        if (!delta) continue;
        // a key leaves with the pose it entered with
        float M[12];
        kf_matrix(delta > 0 ? d.kf_pose[k].data() : d.inc_pose[k].data(), M);
        for (int rep = 0; rep < std::abs(delta); rep++)
            for (int which = 0; which < 2; which++) {
                if (np >= m.max_pieces) { d.inc_ready = false; return false; }
                IncPiece& pc = m.h_pieces[np++];
                pc.which = which; pc.sign = delta > 0 ? 1 : -1; pc.kf = k;
                pc.in_off = which ? d.kf_off_s[k] : d.kf_off_c[k];
                pc.n = which ? d.kf_n_s[k] : d.kf_n_c[k];
                for (int q = 0; q < 12; q++) pc.A[q] = M[q];
                maxn = std::max(maxn, pc.n);
                if (delta > 0) added += pc.n;
            }
        if (delta > 0 && d.inc_mult[k] == 0) d.inc_pose[k] = d.kf_pose[k];
        d.inc_mult[k] = want[k];
    }
    d.inc_nocc_bound = (int)std::min<long long>((long long)d.inc_nocc_bound + added, 0x7fffffff);
    const float leaf[2] = {d.P.mappingCornerLeafSize, d.P.mappingSurfLeafSize};
    if (np) {
        LVI_HIP(hipMemcpyAsync(m.d_pieces, m.h_pieces, sizeof(IncPiece) * (size_t)np, hipMemcpyHostToDevice, cx.stream));
        incmap_apply(cx, m, d.kfPool, np, maxn, leaf);
    }
    // unique active keys for the bounding box (pinned staging: the tail of h_pieces' allocation)
    int na = 0;
    long long tc = 0, ts = 0;
    for (int k = 0; k < nkf; k++) if (want[k] > 0) { m.h_active[na++] = k; tc += (long long)want[k] * d.kf_n_c[k]; ts += (long long)want[k] * d.kf_n_s[k]; }
    if (na) LVI_HIP(hipMemcpyAsync(m.d_active, m.h_active, sizeof(int) * (size_t)na, hipMemcpyHostToDevice, cx.stream));
    incmap_emit(cx, m, na, leaf, d.voxMap.d_grid, d.voxMap.d_nout, d.mapCornerDS, d.mapSurfDS, d.map_cap);
    d.n_map_corner = (int)std::min<long long>(tc, d.map_cap); d.n_map_surf = (int)std::min<long long>(ts, d.map_cap);
    stage_map_index(OneSlot(d).s, cx);
    LVI_HIP(hipMemcpyAsync(m.h_status, m.d_status, sizeof(int), hipMemcpyDeviceToHost, cx.stream));
    LVI_HIP(hipMemcpyAsync(m.h_status + 1, m.d_nocc, sizeof(int) * 2, hipMemcpyDeviceToHost, cx.stream));
    if (forked) { LVI_HIP(hipEventRecord(d.evMap, cx.stream)); d.map_pending = true; }
    // one 12-byte read tells the host whether the device accepted the lists (range, table size, PCL's overflow rule): it
    // waits for the map stream only — the scan-side stages keep running on the main stream
    LVI_HIP(hipStreamSynchronize(cx.stream));
    d.inc_nocc_bound = std::max(m.h_status[1], m.h_status[2]);
    if (m.h_status[0] != 0) { d.inc_ready = false; return false; } // the tables are rebuilt next time; this list goes the full way
    d.inc_ready = true;
    d.have_map = true; d.have_map_raw = false;                     // no fused raw cloud exists in this form
    d.voxMap.bbox_cached = false;
    return true;
}

// f-4.  extractCloud's fuse loop (mapOptimization.cpp:931-957): every listed keyframe cloud through
// transformPointCloud (:347-366) with the key's pose, written at its place in laserCloud{Corner,Surf}FromMap.
// The 3x4 matrices come from the host (pcl::getTransformation evaluated with libm, as in the reference), so the
// fused clouds are bit-identical to the CPU's; blockIdx.y = piece, grid-stride over its points.
__global__ __launch_bounds__(256) void kf_assemble_kernel(const LidarDev::KfSeg* __restrict__ segs, const lvi_pt* __restrict__ pool,
                                                          lvi_pt* __restrict__ outC, lvi_pt* __restrict__ outS)
{
    const LidarDev::KfSeg sg = segs[blockIdx.y];
    const lvi_pt* __restrict__ in = pool + sg.in_off;
    lvi_pt* __restrict__ out = (sg.which ? outS : outC) + sg.out_off;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < sg.n; i += gridDim.x * 256) out[i] = to_map(sg.A, in[i]);
}

void stage_map_assemble(LidarDev& d, const int32_t* keys, int n_keys)
{
    join_map(d);                                    // the previous build may still read the raw buffers
    // h_kfSeg is reused by every call: the previous assembly's H2D copy must have been consumed
    LVI_HIP(hipStreamSynchronize(d.ctx.stream));
    int oc = 0, os = 0, nseg = 0, maxn = 1;
    for (int i = 0; i < n_keys; i++) {
        const int k = keys[i];
        float M[12];
        kf_matrix(d.kf_pose[k].data(), M);
        for (int which = 0; which < 2; which++) {
            LidarDev::KfSeg& sg = d.h_kfSeg[nseg++];
            sg.which = which;
            sg.in_off = which ? d.kf_off_s[k] : d.kf_off_c[k];
            sg.n = which ? d.kf_n_s[k] : d.kf_n_c[k];
            sg.out_off = which ? os : oc;
            for (int q = 0; q < 12; q++) sg.A[q] = M[q];
            (which ? os : oc) += sg.n;
            maxn = std::max(maxn, sg.n);
        }
    }
    if (nseg) {
        LVI_HIP(hipMemcpyAsync(d.d_kfSeg, d.h_kfSeg, sizeof(LidarDev::KfSeg) * (size_t)nseg, hipMemcpyHostToDevice, d.ctx.stream));
        const dim3 grid(std::min(div_up(maxn, 256), 64), nseg);
        LVI_LAUNCH(d.ctx, "kf_assemble", 32.0 * ((double)oc + os), hipLaunchKernelGGL(kf_assemble_kernel, grid, dim3(256), 0, d.ctx.stream,
                                                                                     d.d_kfSeg, d.kfPool, d.mapCornerRaw, d.mapSurfRaw));
    }
    d.n_map_corner = oc; d.n_map_surf = os; d.have_map_raw = true;
    d.voxMap.bbox_cached = false;
}

void set_pose_init(const Slots& sl, const float* p, bool clear_status)
{
    Batch<PoseInitArgs> B;
    for (int z = 0; z < sl.n; z++) {
        B.a[z].dst = sl[z].d_pose_init;
        for (int k = 0; k < 6; k++) B.a[z].t[k] = p[6 * z + k];
        B.a[z].d_status = clear_status ? sl[z].d_status : nullptr;
        B.a[z].st = sl[z].icp; B.a[z].cyc = sl[z].d_icp_cycles; B.a[z].acc = sl[z].icpAcc;
    }
    for (int z = sl.n; z < MAX_BATCH; z++) B.a[z] = B.a[0];
    const Ctx& cx = sl.first().ctx;
    hipLaunchKernelGGL(set_pose_init_kernel, dim3(1, 1, sl.n), dim3(64), 0, cx.stream, B);
    LVI_HIP(hipGetLastError());
}
void set_pose_init(LidarDev& d, const float p[6]) { set_pose_init(OneSlot(d).s, p, false); }

// Features the GN launches' grid is sized for: what the last finished matches of these slots had (+ 1/8, + a block), or the capacity
// while nothing is known.  A hint: the kernel walks on when a scan has more (LVI_GN_GRID=cap: always the capacity).
static int gn_grid_features(const Slots& sl)
{
    const bool full = getenv("LVI_GN_GRID") != nullptr;
    const LidarDev& d0 = sl.first();
    if (const char* e = getenv("LVI_GN_GRID_FEATURES")) { const int f = atoi(e); if (f > 0) return std::min(d0.ext_cap, f); }      // tests: a grid far too small
    int seen = 0;
    for (int z = 0; z < sl.n; z++) {
        const int v = sl[z].h_gn_feat ? *(volatile const int*)sl[z].h_gn_feat : 0;
        if (v <= 0) return d0.ext_cap;                 // a slot that has not reported yet
        seen = std::max(seen, v);
    }
    if (full || seen <= 0) return d0.ext_cap;
    return std::min(d0.ext_cap, seen + seen / 8 + 256);
}

void stage_scan_match_enqueue(const Slots& sl, const lvi_imu_hint* imu, void* d_records, int it_begin, int it_end)
{
    LidarDev& d = sl.first();
    join_map(d);
    Batch<IcpArgs> B;
    double Q = 0;
    for (int z = 0; z < sl.n; z++) {
        IcpArgs a = icp_args(sl[z]);
        a.imu_available = imu ? imu->imu_available : 0;
        a.imu_roll = imu ? imu->imu_roll_init : 0.f;
        a.imu_pitch = imu ? imu->imu_pitch_init : 0.f;
        a.d_record = d_records ? (void*)((char*)d_records + sizeof(lvi_pose_record) * (size_t)z) : nullptr;
        a.have_map = sl[z].have_map ? 1 : 0;
        B.a[z] = a;
        Q += 0.25 * sl[z].n_raw;           // nominal query count for byte accounting only
    }
    for (int z = sl.n; z < MAX_BATCH; z++) B.a[z] = B.a[0];
    const IcpArgs& a = B.a[0];
    const unsigned S = (unsigned)sl.n;
    const Ctx& cx = d.ctx;
    const int it_last = it_end < 0 ? a.max_iters : std::min(it_end, a.max_iters);
    const int ext_cap_l = gn_grid_features(sl);
    for (int it = it_begin; it < it_last; it++) {
        // (the grid covers ext_cap features; the ~1 200 workgroups beyond the actual count exit at once — measured: launching
        // exactly the occupied 360 instead changes nothing)
        // iteration 0 searches the unit ball; later iterations search the (much smaller) ball of the previous neighbours, where
        // the per-lane fixed cost dominates: fewer lanes per feature (d.icp_g1).  The solve and the pose update of an iteration run at
        // the head of the NEXT launch, in every workgroup: ONE launch per Gauss-Newton iteration, no workgroup waits for another.
        const int G1 = it == 0 ? d.icp_g0 : d.icp_g1;
        // a multiple of 8 workgroups: the kernel deals them to the XCDs in contiguous ranges
        // iteration 0 and the two after it search nearly everything (the first corrections move a feature 20 m out by half a metre):
        // 64 features per workgroup, many workgroups; later iterations mostly skip their searches: 256 features per workgroup
        if (it == 0) {
            const dim3 rg((div_up(ext_cap_l, 64) + 7) & ~7, 1, S);
            if (G1 == 8) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<64, true, 8, 8>), rg, dim3(512), 0, cx.stream, B, it));
            else if (G1 == 2) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<64, true, 2, 4>), rg, dim3(128), 0, cx.stream, B, it));
            else if (a.lds_tiles) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<64, true, 4, 4, true>), rg, dim3(256), 0, cx.stream, B, it));
            else LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<64, true, 4, 4>), rg, dim3(256), 0, cx.stream, B, it));
        } else if (it < d.icp_wide_from) {
            const dim3 rg((div_up(ext_cap_l, 64) + 7) & ~7, 1, S);
            if (G1 == 8) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<64, false, 8, 4>), rg, dim3(512), 0, cx.stream, B, it));
            else if (G1 == 2) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<64, false, 2, 4>), rg, dim3(128), 0, cx.stream, B, it));
            else if (a.lds_tiles) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<64, false, 4, 4, true>), rg, dim3(256), 0, cx.stream, B, it));
            else LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<64, false, 4, 4>), rg, dim3(256), 0, cx.stream, B, it));
        } else {
            const dim3 rg((div_up(ext_cap_l, 256) + 7) & ~7, 1, S);
            if (G1 == 8) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<256, false, 8, 4>), rg, dim3(256), 0, cx.stream, B, it));
            else if (G1 == 2) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<256, false, 2, 4>), rg, dim3(256), 0, cx.stream, B, it));
            else if (a.lds_tiles) LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<256, false, 4, 4, true>), rg, dim3(256), 0, cx.stream, B, it));
            else LVI_LAUNCH(cx, "icp_gn", 128.0 * Q, hipLaunchKernelGGL((icp_gn_kernel<256, false, 4, 4>), rg, dim3(256), 0, cx.stream, B, it));
        }
    }
    LVI_LAUNCH(cx, "icp_final", 0, hipLaunchKernelGGL(icp_final_kernel, dim3(1, 1, S), dim3(64), 0, cx.stream, B, it_last));
    if (sl.n == 1 && !d.P.map_on_main_stream) mark_map_deps(d);     // (the GN loop read the index)
}
void stage_scan_match_enqueue(LidarDev& d, const lvi_imu_hint* imu, void* d_record, int it_begin, int it_end) { stage_scan_match_enqueue(OneSlot(d).s, imu, d_record, it_begin, it_end); }

void debug_knn(LidarDev& d, int which, const lvi_pt* d_queries, int nq, int* d_idx, float* d_sqd)
{
    join_map(d);
    hipLaunchKernelGGL(knn_debug_kernel, dim3(div_up(std::max(nq, 1), 256 / KNN_G)), dim3(256), 0, d.ctx.stream,
                       d.grid[which].meta, d.grid[which].cell_start, d.grid[which].sorted, d_queries, nq, d_idx, d_sqd);
    LVI_HIP(hipGetLastError());
}

void debug_residuals(LidarDev& d, int which, const float pose[6])
{
    join_map(d);
    IcpArgs a = icp_args(d);
    hipLaunchKernelGGL(pose_only_kernel, dim3(1), dim3(1), 0, d.ctx.stream, &d.icp->scratch, pose[0], pose[1], pose[2], pose[3], pose[4], pose[5]);
    hipLaunchKernelGGL(residual_debug_kernel, dim3(d.nblk_icp), dim3(ICP_BLOCK), 0, d.ctx.stream, a, which, &d.icp->scratch, d.coeff, d.flag);
    LVI_HIP(hipGetLastError());
}

void transform_cloud(LidarDev& d, const lvi_pt* d_in, int n, const float pose6[6], lvi_pt* d_out)
{
    hipLaunchKernelGGL(pose_only_kernel, dim3(1), dim3(1), 0, d.ctx.stream, &d.icp->scratch, pose6[0], pose6[1], pose6[2], pose6[3], pose6[4], pose6[5]);
    LVI_LAUNCH(d.ctx, "transform_cloud", 32.0 * n, hipLaunchKernelGGL(transform_dev_pose_kernel, dim3(div_up(std::max(n, 1), 256)), dim3(256), 0, d.ctx.stream,
                                                                     d_in, n, &d.icp->scratch, d_out));
}

}  // namespace lvi
