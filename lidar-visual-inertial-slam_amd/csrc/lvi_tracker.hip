// feature_tracker hot path for gfx950 behind the C-ABI:
//   cv::calcOpticalFlowPyrLK(cur, forw, pts, …, Size(21,21), 3)   feature_tracker.cpp:113
//   cv::goodFeaturesToTrack(forw, n_pts, N, 0.01, MIN_DIST, mask)  feature_tracker.cpp:166
//
// LK: the pyramids of the two resident images are built once per image (pyrDown, integer exact).
// One wavefront per feature walks the levels top→0 inside ONE launch: the 24x24 source tile of the
// previous level is staged in LDS, Scharr derivatives are formed on the fly for that tile only
// (OpenCV differentiates the whole level), the 21x21 patch and its gradient live in registers
// (7 pixels per lane), and every Newton step re-stages the 22x22 target tile and reduces the two
// mismatch sums across the wave.  All patch arithmetic is OpenCV's fixed point (W_BITS = 14) with
// exact int64 sums (the aarch64 build of OpenCV, which is what the reference's Jetson runs), so the
// result does not depend on reduction order and matches the CPU restatement bit for bit.
//
// GFTT: min-eigenvalue map (Sobel 3x3 → products → 3x3 box, f64 box accumulation as cv::boxFilter),
// masked max (order-encoded atomicMax), threshold + 3x3 non-maximum test + ordered compaction,
// stable radix sort by value (descending, ties by higher address first), and the sequential
// minimum-distance pick done by one wavefront with ballot-resolved conflicts.
#include <algorithm>

#include "lvi_sort.hpp"

namespace lvi {

namespace {

constexpr int MAX_LEVELS = 8;
constexpr int LK_WIN_MAX = 21;

struct Level { uint8_t* px; int w, h; };
struct Pyr { Level lv[MAX_LEVELS]; int top; };

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

// ------------------------------------------------------------------------------------------- pyrDown
__global__ __launch_bounds__(256) void pyrdown_kernel(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst, int dw, int dh)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= dw || y >= dh) return;
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int sy = reflect101(2 * y + j - 2, sh);
        const uint8_t* row = src + (size_t)sy * sw;
        const int r = row[reflect101(2 * x - 2, sw)] + 4 * row[reflect101(2 * x - 1, sw)] + 6 * row[2 * x < sw ? 2 * x : reflect101(2 * x, sw)]
                    + 4 * row[reflect101(2 * x + 1, sw)] + row[reflect101(2 * x + 2, sw)];
        sum += (j == 0 || j == 4) ? r : (j == 2 ? 6 * r : 4 * r);
    }
    dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
}

// ------------------------------------------------------------------------------------------- LK
struct LkArgs {
    Pyr prev, next;
    const float* prev_xy; float* next_xy; uint8_t* status; float* err;
    float* h_next_xy; uint8_t* h_status; float* h_err;     // host-mapped mirrors of the three results (nullptr: none): get_lk reads them after one wait, no copy commands
    int n, win, max_level, max_count;
    double epsilon; float min_eig;
};

#define LVI_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
__device__ __forceinline__ int cv_floor(float v) { const int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }

// ---------------------------------------------------------------------------------------------
// f-2  cv::CLAHE::apply, CV_8UC1 (OpenCV imgproc/src/clahe.cpp).  clahe_lut: one workgroup per tile — LDS
// histogram, clip + redistribution evaluated per bin (the serial residual loop hits bins 0, step, 2·step, …),
// inclusive scan, lut = saturate(cvRound(sum * lutScale)).  clahe_interp: one thread per pixel, the f32 blend of
// the four surrounding tile LUTs in the reference's operation order (the library is built -ffp-contract=off).
// ---------------------------------------------------------------------------------------------
struct ClaheArgs {
    const uint8_t* src; uint8_t* dst; uint8_t* lut;
    int W, H, tilesX, tilesY, tw, th, clipLimit; float lutScale;
};

__global__ __launch_bounds__(1024) void clahe_lut_kernel(ClaheArgs a)
{
    // 1024 threads per tile, four pixels per load, one histogram per wavefront (a tile of the 1280 x 720 frame is 14 400 pixels: 256
    // threads reading bytes one by one into ONE histogram were 32 us of the node's frame — 64 workgroups, 56 dependent rounds each)
    __shared__ int whist[16][256];
    __shared__ int ws[17];
    const int k = blockIdx.x, tx = k % a.tilesX, ty = k / a.tilesX, tid = threadIdx.x, wv = tid >> 6;
    for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
    __syncthreads();
    const int area = a.tw * a.th;
    const int x0 = tx * a.tw, y0 = ty * a.th;
    const bool vec = (a.tw & 3) == 0 && (a.W & 3) == 0 && x0 + a.tw <= a.W && y0 + a.th <= a.H;     // (x0 is then a multiple of 4 as well)
    if (vec) {
        const int qw = a.tw >> 2, nq = qw * a.th;
        constexpr int NL = 4;
        for (int q0 = tid; q0 < nq; q0 += NL * 1024) {
            unsigned v[NL];
#pragma unroll
            for (int u = 0; u < NL; u++) {
                const int q = min(q0 + u * 1024, nq - 1);
                v[u] = *reinterpret_cast<const unsigned*>(a.src + (size_t)(y0 + q / qw) * a.W + x0 + 4 * (q % qw));
            }
#pragma unroll
            for (int u = 0; u < NL; u++) {
                if (q0 + u * 1024 >= nq) continue;
                atomicAdd(&whist[wv][v[u] & 255u], 1); atomicAdd(&whist[wv][(v[u] >> 8) & 255u], 1);
                atomicAdd(&whist[wv][(v[u] >> 16) & 255u], 1); atomicAdd(&whist[wv][v[u] >> 24], 1);
            }
        }
    } else {
        for (int idx = tid; idx < area; idx += 1024) {
            const int x = x0 + idx % a.tw, y = y0 + idx / a.tw;
            // right / bottom extension of images that are not a multiple of the tile grid: BORDER_REFLECT_101
            const int sx = x < a.W ? x : reflect101(x, a.W), sy = y < a.H ? y : reflect101(y, a.H);
            atomicAdd(&whist[wv][a.src[(size_t)sy * a.W + sx]], 1);
        }
    }
    __syncthreads();
    // bins live in the first 256 threads; the others contribute zeros to the two scans (every thread reaches every barrier)
    int hv = 0;
    if (tid < 256) {
#pragma unroll
        for (int q = 0; q < 16; q++) hv += whist[q][tid];
    }
    if (a.clipLimit > 0) {
        const int excess = max(hv - a.clipLimit, 0);
        hv = min(hv, a.clipLimit);
        int clipped;
        (void)block_excl_scan<1024>(excess, ws, &clipped);
        const int redistBatch = clipped / 256;
        const int residual = clipped - redistBatch * 256;
        if (tid < 256) {
            hv += redistBatch;
            if (residual != 0) {
                const int step = max(256 / residual, 1);
                if (tid % step == 0 && tid / step < residual) hv++;
            }
        }
    }
    int tot;
    const int sum = block_excl_scan<1024>(hv, ws, &tot) + hv;
    const int v = cv_round((float)sum * a.lutScale);
    if (tid >= 256) return;
    a.lut[(size_t)k * 256 + tid] = (uint8_t)min(max(v, 0), 255);
}

__global__ __launch_bounds__(256) void clahe_interp_kernel(ClaheArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const float inv_tw = 1.0f / a.tw, inv_th = 1.0f / a.th;
    const float tyf = (float)y * inv_th - 0.5f;
    int ty1 = cv_floor(tyf), ty2 = ty1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0); ty2 = min(ty2, a.tilesY - 1);
    const float txf = (float)x * inv_tw - 0.5f;
    int tx1 = cv_floor(txf), tx2 = tx1 + 1;
    const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0); tx2 = min(tx2, a.tilesX - 1);
    const int srcVal = a.src[(size_t)y * a.W + x];
    const uint8_t* p1 = a.lut + (size_t)ty1 * a.tilesX * 256;
    const uint8_t* p2 = a.lut + (size_t)ty2 * a.tilesX * 256;
    const int ind1 = tx1 * 256 + srcVal, ind2 = tx2 * 256 + srcVal;
    const float res = ((float)p1[ind1] * xa1 + (float)p1[ind2] * xa) * ya1 + ((float)p2[ind1] * xa1 + (float)p2[ind2] * xa) * ya;
    a.dst[(size_t)y * a.W + x] = (uint8_t)min(max(cv_round(res), 0), 255);
}

// ---------------------------------------------------------------------------------------------
// f-3  CataCamera::liftProjective with the 8-step recursive distortion model + (b.x/b.z, b.y/b.z) → Point2f
// (CataCamera.cc:556-626, 766-783; feature_tracker.cpp:306-309).  All in double, one thread per point.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void mei_distortion(const lvi_mei_params& c, double pux, double puy, double& dux, double& duy)
{
    const double mx2_u = pux * pux, my2_u = puy * puy, mxy_u = pux * puy;
    const double rho2_u = mx2_u + my2_u;
    const double rad_dist_u = c.k1 * rho2_u + c.k2 * rho2_u * rho2_u;
    dux = pux * rad_dist_u + 2.0 * c.p1 * mxy_u + c.p2 * (rho2_u + 2.0 * mx2_u);
    duy = puy * rad_dist_u + 2.0 * c.p2 * mxy_u + c.p1 * (rho2_u + 2.0 * my2_u);
}

__global__ __launch_bounds__(64) void mei_undistort_kernel(lvi_mei_params c, const float* __restrict__ xy, int n, float* __restrict__ out, const int* __restrict__ n_dev = nullptr)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);                  // (the count of a device-side concatenation)
    if (i >= n) return;
    const double inv_K11 = 1.0 / c.gamma1, inv_K13 = -c.u0 / c.gamma1, inv_K22 = 1.0 / c.gamma2, inv_K23 = -c.v0 / c.gamma2;
    const bool noDistortion = c.k1 == 0.0 && c.k2 == 0.0 && c.p1 == 0.0 && c.p2 == 0.0;
    const double mx_d = inv_K11 * (double)xy[2 * i] + inv_K13, my_d = inv_K22 * (double)xy[2 * i + 1] + inv_K23;
    double mx_u = mx_d, my_u = my_d;
    if (!noDistortion) {
        double dux, duy;
        mei_distortion(c, mx_d, my_d, dux, duy);
        mx_u = mx_d - dux; my_u = my_d - duy;
        for (int it = 1; it < 8; ++it) { mei_distortion(c, mx_u, my_u, dux, duy); mx_u = mx_d - dux; my_u = my_d - duy; }
    }
    double bz;
    if (c.xi == 1.0) bz = (1.0 - mx_u * mx_u - my_u * my_u) / 2.0;
    else { const double rho2_d = mx_u * mx_u + my_u * my_u; bz = 1.0 - c.xi * (rho2_d + 1.0) / (c.xi + sqrt(1.0 + (1.0 - c.xi * c.xi) * rho2_d)); }
    out[2 * i] = (float)(mx_u / bz); out[2 * i + 1] = (float)(my_u / bz);
}

// setMask (feature_tracker.cpp:36-69): mask = 255, then cv::circle(mask, pt, MIN_DIST, 0, -1) around every kept point.  The
// filled circle is OpenCV's midpoint raster (FillCircle): row cy +- dy spans cx +- dx and row cy +- dx spans cx +- dy for the
// (dx, dy) the integer error recurrence visits; hw[j] = the widest of the spans of row offset j.  One thread per (circle, row).
struct CircleArgs { uint8_t* mask; int w, h; const float* centers; int n; int radius; };
__global__ __launch_bounds__(256) void mask_fill_kernel(CircleArgs a)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i + 16 <= (size_t)a.w * a.h) *reinterpret_cast<uint4*>(a.mask + i) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    else for (size_t k = i; k < (size_t)a.w * a.h; k++) a.mask[k] = 255;
}
__global__ __launch_bounds__(64) void mask_circles_kernel(CircleArgs a)
{
    __shared__ int hw[128];
    const int r = a.radius;
    if (threadIdx.x == 0) {
        for (int j = 0; j <= r; j++) hw[j] = -1;
        int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1;
        while (dx >= dy) {
            hw[dy] = max(hw[dy], dx); hw[dx] = max(hw[dx], dy);
            dy++;
            err += plus; plus += 2;
            const int m = (err <= 0) - 1;
            err -= minus & m; dx += m; minus -= m & 2;
        }
    }
    __syncthreads();
    const int c = blockIdx.x;
    if (c >= a.n) return;
    const int cx = (int)rintf(a.centers[2 * c]), cy = (int)rintf(a.centers[2 * c + 1]);      // Mat::at(Point2f) / cv::circle round the centre (cvRound)
    for (int j = threadIdx.x; j <= 2 * r; j += 64) {
        const int off = j - r, y = cy + off, half = hw[off < 0 ? -off : off];
        if (y < 0 || y >= a.h || half < 0) continue;
        const int x0 = max(cx - half, 0), x1 = min(cx + half, a.w - 1);
        for (int x = x0; x <= x1; x++) a.mask[(size_t)y * a.w + x] = 0;
    }
}

// undistortedPoints over [the kept points (host) ; the corners goodFeaturesToTrack just found (device)]: cur_pts of the next frame
__global__ __launch_bounds__(64) void frame_concat_kernel(const float* __restrict__ kept, int n_kept, const float* __restrict__ found, const int* __restrict__ n_found,
                                                          const int* __restrict__ n_cand, int with_gftt, int cap, float* __restrict__ all_xy, int* __restrict__ d_hdr,
                                                          int* __restrict__ h_hdr, float* __restrict__ h_new)
{
    // kept: the caller's points in pinned host memory (read in place); h_hdr / h_new: the frame's result block in pinned host memory —
    // {n_new (or the pick kernel's negative code), n_cand, n_all}, then the new corners: the frame end is kernels and one wait
    const int nraw = with_gftt ? *n_found : 0;
    const int nf = min(max(nraw, 0), cap);
    const int n = min(n_kept + nf, cap);
    for (int i = threadIdx.x; i < n; i += 64) {
        const float* src = i < n_kept ? kept + 2 * i : found + 2 * (i - n_kept);
        const float x = src[0], y = src[1];
        all_xy[2 * i] = x; all_xy[2 * i + 1] = y;
        if (i >= n_kept) { h_new[2 * (i - n_kept)] = x; h_new[2 * (i - n_kept) + 1] = y; }
    }
    if (threadIdx.x == 0) { d_hdr[2] = n; h_hdr[0] = nraw; h_hdr[1] = with_gftt ? *n_cand : 0; h_hdr[2] = n; }
}
__global__ __launch_bounds__(64) void lk_kernel(LkArgs a)
{
    constexpr int TW = LK_WIN_MAX + 3;             // 24: source tile (window + 1 for bilinear + 1 on each side for Scharr)
    constexpr int DW = LK_WIN_MAX + 1;             // 22: derivative / target tile
    constexpr int NPX = (LK_WIN_MAX * LK_WIN_MAX + 63) / 64;   // 7 window pixels per lane
    __shared__ uint8_t sI[TW * TW];
    __shared__ short sD[DW * DW * 2];
    // target image: a region of JW x JW pixels around the window stays in LDS across the iterations of a level (the window moves
    // by a fraction of a pixel per iteration: reloading its 22 x 22 tile from L2 every iteration was most of the kernel's 80 us)
    constexpr int JR = 8, JW = DW + 2 * JR;
    __shared__ uint8_t sJ[JW * JW];
    int jx0 = 0, jy0 = 0; bool jvalid = false;     // absolute coordinates of sJ[0] in the level (uniform over the wavefront)
    const int f = blockIdx.x, l = threadIdx.x;
    if (f >= a.n) return;
    const int win = a.win;
    const float halfWin = (win - 1) * 0.5f;
    // window pixels of this lane, (row, column), once: `win` is a run-time value and an integer division costs ~40 instructions
    // (lanes beyond the window take pixel (0, 0) and are masked out: no branch inside the pixel loops, so that the LDS reads
    // of all seven pixels are in flight together — one wavefront per workgroup has nothing else to hide their latency)
    int wy[NPX], wx[NPX]; bool wv[NPX];
#pragma unroll
    for (int j = 0; j < NPX; j++) { const int p = l + 64 * j; wv[j] = p < win * win; wy[j] = wv[j] ? p / win : 0; wx[j] = wv[j] ? p - wy[j] * win : 0; }
    const float px0 = a.prev_xy[2 * f], py0 = a.prev_xy[2 * f + 1];
    float outx = 0.f, outy = 0.f, errv = 0.f;
    bool st = true;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);

    for (int level = a.max_level; level >= 0; level--) {
        const Level I = a.prev.lv[level], J = a.next.lv[level];
        jvalid = false;
        // make the window [inx, inx + DW) x [iny, iny + DW) of the target level resident; returns the offset of its corner in sJ
        constexpr int NJ = (JW * JW + 63) / 64;
        // all loads of a lane are issued before the first one is stored (a loop of load, wait, store was 23 round trips to L2)
        auto region_issue = [&](int x0, int y0, uint8_t (&rj)[NJ]) {
            if (x0 >= 0 && y0 >= 0 && x0 + JW <= J.w && y0 + JW <= J.h) {                  // interior: no border arithmetic
                const uint8_t* __restrict__ src = J.px + (size_t)y0 * J.w + x0;
#pragma unroll
                for (int k = 0; k < NJ; k++) { const int t = l + 64 * k, ty = t / JW, tx = t - ty * JW; rj[k] = t < JW * JW ? src[ty * J.w + tx] : (uint8_t)0; }
            } else {
#pragma unroll
                for (int k = 0; k < NJ; k++) {
                    const int t = l + 64 * k, ty = t / JW, tx = t - ty * JW;
                    rj[k] = t < JW * JW ? J.px[(size_t)reflect101(y0 + ty, J.h) * J.w + reflect101(x0 + tx, J.w)] : (uint8_t)0;
                }
            }
        };
        auto region_commit = [&](int x0, int y0, const uint8_t (&rj)[NJ]) {
#pragma unroll
            for (int k = 0; k < NJ; k++) { const int t = l + 64 * k; if (t < JW * JW) sJ[t] = rj[k]; }
            jx0 = x0; jy0 = y0; jvalid = true;
        };
        // make the window [inx, inx + DW) x [iny, iny + DW) of the target level resident; returns the offset of its corner in sJ
        auto target_window = [&](int inx, int iny) {
            if (!(jvalid && inx >= jx0 && inx + DW <= jx0 + JW && iny >= jy0 && iny + DW <= jy0 + JW)) {
                __syncthreads();
                uint8_t rj[NJ];
                region_issue(inx - JR, iny - JR, rj);
                region_commit(inx - JR, iny - JR, rj);
                __syncthreads();
            }
            return (iny - jy0) * JW + (inx - jx0);
        };
        float prevx = px0 * (float)(1. / (1 << level)), prevy = py0 * (float)(1. / (1 << level));
        float nextx, nexty;
        if (level == a.max_level) { nextx = prevx; nexty = prevy; }
        else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;
        prevx -= halfWin; prevy -= halfWin;
        const int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
        if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
            if (level == 0) { st = false; errv = 0.f; }
            continue;
        }
        // ---- stage the source tile (REFLECT_101 border of the pyramid level) and its Scharr derivatives
        __syncthreads();
        {
            constexpr int NI = (TW * TW + 63) / 64;
            uint8_t ri[NI];
            if (ipx >= 1 && ipy >= 1 && ipx - 1 + TW <= I.w && ipy - 1 + TW <= I.h) {      // interior: no border arithmetic
                const uint8_t* __restrict__ src = I.px + (size_t)(ipy - 1) * I.w + (ipx - 1);
#pragma unroll
                for (int k = 0; k < NI; k++) { const int t = l + 64 * k, ty = t / TW, tx = t - ty * TW; ri[k] = t < TW * TW ? src[ty * I.w + tx] : (uint8_t)0; }
            } else {
#pragma unroll
                for (int k = 0; k < NI; k++) {
                    const int t = l + 64 * k, ty = t / TW, tx = t - ty * TW;
                    ri[k] = t < TW * TW ? I.px[(size_t)reflect101(ipy - 1 + ty, I.h) * I.w + reflect101(ipx - 1 + tx, I.w)] : (uint8_t)0;
                }
            }
#pragma unroll
            for (int k = 0; k < NI; k++) { const int t = l + 64 * k; if (t < TW * TW) sI[t] = ri[k]; }
        }
        __syncthreads();
        {
            constexpr int ND = (DW * DW + 63) / 64;
            short gxs[ND], gys[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const int t = min(l + 64 * k, DW * DW - 1);
                const int dy_ = t / DW, dx_ = t - dy_ * DW;
                const int X = ipx + dx_, Y = ipy + dy_;
                const uint8_t* c = &sI[(dy_ + 1) * TW + (dx_ + 1)];
                const int p00 = c[-TW - 1], p01 = c[-TW], p02 = c[-TW + 1], p10 = c[-1], p12 = c[1], p20 = c[TW - 1], p21 = c[TW], p22 = c[TW + 1];
                const bool in = X >= 0 && X < I.w && Y >= 0 && Y < I.h;             // the derivative image is zero outside the level
                gxs[k] = in ? (short)(((p02 + p22) * 3 + p12 * 10) - ((p00 + p20) * 3 + p10 * 10)) : (short)0;
                gys[k] = in ? (short)(((p20 - p00) + (p22 - p02)) * 3 + (p21 - p01) * 10) : (short)0;
            }
#pragma unroll
            for (int k = 0; k < ND; k++) { const int t = l + 64 * k; if (t < DW * DW) { sD[2 * t] = gxs[k]; sD[2 * t + 1] = gys[k]; } }
        }
        __syncthreads();
        float aa = prevx - ipx, bb = prevy - ipy;
        int iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
        int iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
        int iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        short Iw[NPX], Ix[NPX], Iy[NPX];
        long long sA11 = 0, sA12 = 0, sA22 = 0;
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            const int y = wy[j], x = wx[j];
            const uint8_t* s = &sI[(y + 1) * TW + (x + 1)];
            const short* d = &sD[2 * (y * DW + x)];
            const int ival = wv[j] ? LVI_DESCALE(s[0] * iw00 + s[1] * iw01 + s[TW] * iw10 + s[TW + 1] * iw11, W_BITS - 5) : 0;
            const int ixval = wv[j] ? LVI_DESCALE(d[0] * iw00 + d[2] * iw01 + d[2 * DW] * iw10 + d[2 * DW + 2] * iw11, W_BITS) : 0;
            const int iyval = wv[j] ? LVI_DESCALE(d[1] * iw00 + d[3] * iw01 + d[2 * DW + 1] * iw10 + d[2 * DW + 3] * iw11, W_BITS) : 0;
            Iw[j] = (short)ival; Ix[j] = (short)ixval; Iy[j] = (short)iyval;
            sA11 += ixval * ixval; sA12 += ixval * iyval; sA22 += iyval * iyval;
        }
        sA11 = wave_sum_i64(sA11); sA12 = wave_sum_i64(sA12); sA22 = wave_sum_i64(sA22);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
        if (minEig < a.min_eig || D < 1.1920929e-7f) {
            if (level == 0) st = false;
            continue;
        }
        D = 1.f / D;
        nextx -= halfWin; nexty -= halfWin;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; j++) {
            const int inx = cv_floor(nextx), iny = cv_floor(nexty);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                if (level == 0) st = false;
                break;
            }
            const int jo = target_window(inx, iny);
            aa = nextx - inx; bb = nexty - iny;
            iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            long long ib1 = 0, ib2 = 0;
#pragma unroll
            for (int q = 0; q < NPX; q++) {
                const uint8_t* s = &sJ[jo + wy[q] * JW + wx[q]];          // a lane beyond the window reads pixel (0, 0): its Ix = Iy = 0
                const int diff = LVI_DESCALE(s[0] * iw00 + s[1] * iw01 + s[JW] * iw10 + s[JW + 1] * iw11, W_BITS - 5) - Iw[q];
                ib1 += diff * Ix[q]; ib2 += diff * Iy[q];
            }
            ib1 = wave_sum_i64(ib1); ib2 = wave_sum_i64(ib2);
            const float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
            const float dx = (float)((A12 * b2 - A22 * b1) * D);
            const float dy = (float)((A12 * b1 - A11 * b2) * D);
            nextx += dx; nexty += dy;
            outx = nextx + halfWin; outy = nexty + halfWin;
            if ((double)dx * dx + (double)dy * dy <= a.epsilon) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01 && fabsf(dy + pdy) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (st && level == 0) {
            const float nx = outx - halfWin, ny = outy - halfWin;
            const int inx = cv_floor(nx), iny = cv_floor(ny);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { st = false; continue; }
            const int jo = target_window(inx, iny);
            aa = nx - inx; bb = ny - iny;
            iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int esum = 0;                                  // |diff| are integers; the f32 sum of OpenCV is exact (< 2^24)
#pragma unroll
            for (int q = 0; q < NPX; q++) {
                const uint8_t* s = &sJ[jo + wy[q] * JW + wx[q]];
                const int diff = LVI_DESCALE(s[0] * iw00 + s[1] * iw01 + s[JW] * iw10 + s[JW + 1] * iw11, W_BITS - 5) - Iw[q];
                esum += wv[q] ? abs(diff) : 0;
            }
            esum = wave_sum(esum);
            errv = (float)esum * 1.f / (32 * win * win);
        }
    }
    if (l == 0) {
        a.next_xy[2 * f] = outx; a.next_xy[2 * f + 1] = outy;
        a.status[f] = st ? 1 : 0; a.err[f] = errv;
        if (a.h_next_xy) { a.h_next_xy[2 * f] = outx; a.h_next_xy[2 * f + 1] = outy; a.h_status[f] = st ? 1 : 0; a.h_err[f] = errv; }
    }
}

// ------------------------------------------------------------------------------------------- GFTT
struct GfttArgs {
    const uint8_t* img; const uint8_t* mask; int w, h;
    float* eig; unsigned* maxord; float* thr;
    unsigned* maxPartial;            // [workgroups of mineig_kernel]
    int* blockCnt; int* total;
    unsigned *keysA, *valsA;
    int *d_n, *d_nbits;
    double quality;
};

__device__ __forceinline__ void sobel_at(const uint8_t* img, int w, int h, int x, int y, float k0, float k1, float& Dx, float& Dy)
{
    // evaluated at in-image (x,y); neighbours through REFLECT_101
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w), ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    const uint8_t *r0 = img + (size_t)ym * w, *r1 = img + (size_t)y * w, *r2 = img + (size_t)yp * w;
    const float a0 = r0[xm], a1 = r0[x], a2 = r0[xp], b0 = r1[xm], b2 = r1[xp], c0 = r2[xm], c1 = r2[x], c2 = r2[xp];
    // dx: row [-1 0 1] exact, column [1 2 1]*scale:  (S0 + S2)*k1 + S1*k0
    const float S0 = a2 - a0, S1 = b2 - b0, S2 = c2 - c0;
    const float u = (S0 + S2) * k1, v = S1 * k0;
    Dx = u + v;
    // dy: row [1 2 1]*scale: S[0]*k0 + (S[-1]+S[1])*k1, column [-1 0 1]
    const float t0 = a1 * k0, t1 = (a0 + a2) * k1, rs0 = t0 + t1;
    const float t2 = c1 * k0, t3 = (c0 + c2) * k1, rs2 = t2 + t3;
    Dy = rs2 - rs0;
}

// Tile of 32 x 8 output pixels: the Sobel products of the 34 x 10 pixels under the 3x3 box are computed ONCE into
// LDS (the first version recomputed them nine times per pixel), then every thread adds its nine in the reference's
// order (rows, then columns, in double).  The masked maximum goes to one partial per workgroup; gftt_thr folds them
// (14 400 wavefronts hitting one atomicMax address cost ~0.16 ms: same-address atomics serialise).
__global__ __launch_bounds__(256) void mineig_kernel(GfttArgs a)
{
    constexpr int TW = 34, TH = 10;
    __shared__ float sxx[TH][TW], sxy[TH][TW], syy[TH][TW];
    __shared__ unsigned smax[4];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8;
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k1 = (float)(1.0 * scale), k0 = (float)(2.0 * scale);
    for (int e = threadIdx.x; e < TW * TH; e += 256) {
        const int ex = e % TW, ey = e / TW;
        // the box reads pixel reflect101(x + i), reflect101(y + j); Sobel is evaluated at that in-image pixel
        const int px = reflect101(x0 - 1 + ex, a.w), py = reflect101(y0 - 1 + ey, a.h);
        float dx, dy;
        sobel_at(a.img, a.w, a.h, px, py, k0, k1, dx, dy);
        sxx[ey][ex] = dx * dx; sxy[ey][ex] = dx * dy; syy[ey][ex] = dy * dy;
    }
    __syncthreads();
    const int x = x0 + tx, y = y0 + ty;
    float val = 0.f; const bool in = (x < a.w && y < a.h);
    if (in) {
        double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) { s0 += sxx[ty + j][tx + i]; s1 += sxy[ty + j][tx + i]; s2 += syy[ty + j][tx + i]; }
        const float A = (float)s0 * 0.5f, B = (float)s1, C = (float)s2 * 0.5f;
        const float t = (A - C) * (A - C), u = B * B;
        val = (A + C) - sqrtf(t + u);
        a.eig[(size_t)y * a.w + x] = val;
    }
    // masked maximum (minMaxLoc with mask)
    const bool counted = in && (!a.mask || a.mask[(size_t)y * a.w + x]);
    unsigned o = counted ? f2ord(val) : 0u;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { const unsigned t = __shfl_xor(o, s, 64); o = t > o ? t : o; }
    if (lane_id() == 0) smax[wave_id()] = o;
    __syncthreads();
    if (threadIdx.x == 0) a.maxPartial[blockIdx.y * gridDim.x + blockIdx.x] = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
}

// quality threshold = max over the (masked) min-eigenvalue image x qualityLevel (featureselect.cpp): the fold of mineig's per-workgroup
// maxima.  Every workgroup of gftt_count folds the few thousand partial maxima itself (a separate one-workgroup launch was 4.6 us of
// the node's frame); its first workgroup leaves the result for gftt_emit and the debug views.
__device__ __forceinline__ float gftt_threshold(const GfttArgs& a, int npartial, unsigned* smax)
{
    unsigned o = 0u;
    for (int i = threadIdx.x; i < npartial; i += 256) o = max(o, a.maxPartial[i]);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { const unsigned t = __shfl_xor(o, s, 64); o = t > o ? t : o; }
    if (lane_id() == 0) smax[wave_id()] = o;
    __syncthreads();
    o = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    const double maxVal = o ? (double)ord2f(o) : 0.0;
    const float thr = (float)(maxVal * a.quality);
    if (blockIdx.x == 0 && threadIdx.x == 0) { *a.maxord = o; *a.thr = thr; }
    return thr;
}

__device__ __forceinline__ bool gftt_is_cand(const GfttArgs& a, int x, int y, float thr)
{
    // the 3 x 3 neighbourhood and the mask byte are loaded unconditionally (coordinates clamped into the interior) and tested
    // afterwards: with the tests in between, a pixel was up to ten round trips in a row
    const int xc = min(max(x, 1), a.w - 2), yc = min(max(y, 1), a.h - 2);
    const float* e = a.eig + (size_t)yc * a.w + xc;
    float nv[9];
#pragma unroll
    for (int j = -1; j <= 1; j++)
#pragma unroll
        for (int i = -1; i <= 1; i++) nv[(j + 1) * 3 + (i + 1)] = e[j * a.w + i];
    const uint8_t mk = a.mask ? a.mask[(size_t)yc * a.w + xc] : (uint8_t)1;
    if (x < 1 || y < 1 || x >= a.w - 1 || y >= a.h - 1) return false;
    const float val = nv[4];
    if (!(val > thr) || val == 0.f || !mk) return false;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 9; k++) ok = ok && !(nv[k] > thr && nv[k] > val);
    return ok;
}

constexpr int CAND_TILE = 1024;
__global__ __launch_bounds__(256) void gftt_count_kernel(GfttArgs a, int npartial)
{
    __shared__ unsigned smax[4];
    const float thr = gftt_threshold(a, npartial, smax);
    const int base = blockIdx.x * CAND_TILE;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int p = base + threadIdx.x * 4 + j; if (p < a.w * a.h && gftt_is_cand(a, p % a.w, p / a.w, thr)) c++; }
    __shared__ int ws[8];
    int tot;
    block_excl_scan<256>(c, ws, &tot);
    if (threadIdx.x == 0) a.blockCnt[blockIdx.x] = tot;
}
// ordered compaction; the prefix over the (<= a few thousand) per-tile counts is folded by every workgroup itself (the one-workgroup
// scan launch in between was 4.8 us of the node's frame)
__global__ __launch_bounds__(256) void gftt_emit_kernel(GfttArgs a, int nblk)
{
    const float thr = *a.thr;
    __shared__ int ws[8];
    int bef = 0, all = 0;
    for (int i = threadIdx.x; i < nblk; i += 256) { const int v = a.blockCnt[i]; all += v; bef += i < (int)blockIdx.x ? v : 0; }
    int total, before;
    (void)block_excl_scan<256>(all, ws, &total);
    (void)block_excl_scan<256>(bef, ws, &before);
    if (blockIdx.x == 0 && threadIdx.x == 0) { *a.total = total; a.d_n[0] = total; a.d_nbits[0] = 32; }
    const int base = blockIdx.x * CAND_TILE;
    bool h[4]; int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int p = base + threadIdx.x * 4 + j; h[j] = (p < a.w * a.h) && gftt_is_cand(a, p % a.w, p / a.w, thr); c += h[j]; }
    int r = before + block_excl_scan<256>(c, ws, nullptr);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!h[j]) continue;
        const int p = base + threadIdx.x * 4 + j;
        // emit in DESCENDING address order: the stable sort by value then keeps "higher address first" among ties
        const int pos = total - 1 - r++;
        a.keysA[pos] = ~__float_as_uint(a.eig[p]);          // eig > 0 here: ascending ~bits = descending value
        a.valsA[pos] = (unsigned)p;
    }
}

struct PickArgs {
    const unsigned *keysA;
    const unsigned *valsA, *valsB; const int* d_nbits; const int* total;
    int w, h, max_corners, cap; double min_dist;
    float* out_xy; int* out_n; int* ncand;
    long long* dbg;                  // [8] phase stamps of the LDS form (diagnostics; may be null)
};

// sequential minimum-distance selection (featureselect.cpp), one wavefront, ballot-resolved.  A candidate is compared with
// the accepted corners of the 3 x 3 grid cells around it (cell = min_dist, as OpenCV's own grid): the cells keep up to four
// accepted indices each in LDS (corners at least min_dist apart: two fit a cell), so a candidate reads a handful of entries
// instead of walking all accepted corners; a fifth corner in one cell, or a grid beyond the LDS budget, falls back to that walk.
// SORTLDS = false: the candidates arrive sorted (12-launch radix sort), one wavefront picks.
// SORTLDS = true: ONE workgroup of 1024 threads.  The greedy pick consumes candidates in descending value order and stops at the
// quota (MAX_CNT - tracked: tens of corners), so only the strongest few thousand are ever looked at: the workgroup takes a
// 1024-bin histogram of the values, cuts a band of at most GFTT_BAND candidates off its top, sorts THAT band in LDS (bitonic;
// keys = ~value bits, then ~address: descending value, higher address first among equal values — cv::goodFeaturesToTrack's
// greaterThanPtr order — and unique), lets its first wavefront pick from it, and goes on to the next band only if the quota is
// not met.  One launch instead of thirteen; any number of candidates.  A value bin that alone exceeds the band (thousands of
// exactly equal values) or an accepted list beyond the LDS tables reports -2: the host runs the radix form for that frame.
constexpr int GFTT_BAND = 1024;                   // one candidate per thread: the band is ordered by counting, for every key, the keys below it
template <bool SORTLDS>
__global__ __launch_bounds__(SORTLDS ? 1024 : 64) void gftt_pick_kernel(PickArgs a)
{
    constexpr int ACC_MAX = SORTLDS ? 2048 : 4096, GRID_MAX = SORTLDS ? 4096 : 8192, CELL_CAP = 4;
    constexpr int NT = SORTLDS ? 1024 : 64;
    __shared__ short ax[ACC_MAX], ay[ACC_MAX], acx[ACC_MAX], acy[ACC_MAX];      // accepted corners and their grid cells (no division in the inner loop)
    __shared__ unsigned gcnt[GRID_MAX];                                          // (words: the accepted corners of a batch enter their cells together, through atomics)
    __shared__ unsigned short gent[GRID_MAX * CELL_CAP];
    __shared__ unsigned long long band[SORTLDS ? GFTT_BAND : 1];
    __shared__ unsigned hist[SORTLDS ? 1024 : 1], hsub[SORTLDS ? 8 * 1024 : 1];      // hsub: eight copies (lane & 7) — weak candidates crowd the last bins
    __shared__ int ws[NT / 64 + 1];
    __shared__ unsigned s_kmin, s_kmax;
    __shared__ int s_lo, s_hi, s_state;                   // the band's bin range [lo, hi); 0 = go on | 1 = finished | 2 = redo in the radix form
    const int total = *a.total;
    const unsigned* vals = rs_result_in_B(a.d_nbits[0]) ? a.valsB : a.valsA;
    const int l = threadIdx.x;
    if (l == 0) { *a.ncand = total; s_state = 0; s_lo = 0; s_hi = 0; s_kmin = 0xFFFFFFFFu; s_kmax = 0u; }
    const bool filter = a.min_dist >= 1.0;
    const int cell = filter ? (int)rint(a.min_dist) : 1;
    const double md2 = a.min_dist * a.min_dist;
    const int gw = (a.w + cell - 1) / cell, gh = (a.h + cell - 1) / cell;
    bool grid = filter && gw * gh <= GRID_MAX;
    if (grid) for (int c = l; c < gw * gh; c += NT) gcnt[c] = 0;
    if (SORTLDS) for (int b = l; b < 8 * 1024; b += NT) hsub[b] = 0u;
    __syncthreads();
    int shift = 0;
    constexpr int KPT = SORTLDS ? 16 : 1;
    unsigned kreg[KPT];
    long long tq[6] = {0, 0, 0, 0, 0, 0}; int nbands = 0;
    if (SORTLDS && l == 0) tq[0] = clock64();
    if (SORTLDS) {
        // range of the value keys, then the histogram (keys ascend as values descend: bin 0 holds the strongest)
        // (the first KPT keys of a thread stay in registers for all three passes — range, histogram, band placement; one round of
        // loads, all in flight together: 14 k candidates were 14 dependent round trips per pass)
        unsigned kmn = 0xFFFFFFFFu, kmx = 0u;
#pragma unroll
        for (int j = 0; j < KPT; j++) kreg[j] = a.keysA[min(l + j * NT, max(total - 1, 0))];
#pragma unroll
        for (int j = 0; j < KPT; j++) if (l + j * NT < total) { kmn = min(kmn, kreg[j]); kmx = max(kmx, kreg[j]); }
        for (int i = l + KPT * NT; i < total; i += NT) { const unsigned k = a.keysA[i]; kmn = min(kmn, k); kmx = max(kmx, k); }
        atomicMin(&s_kmin, kmn); atomicMax(&s_kmax, kmx);
        __syncthreads();
        while (((s_kmax - s_kmin) >> shift) >= 1024u) shift++;
#pragma unroll
        for (int j = 0; j < KPT; j++) if (l + j * NT < total) atomicAdd(&hsub[(l & 7) * 1024 + ((kreg[j] - s_kmin) >> shift)], 1u);
        for (int i = l + KPT * NT; i < total; i += NT) atomicAdd(&hsub[(l & 7) * 1024 + ((a.keysA[i] - s_kmin) >> shift)], 1u);
        __syncthreads();
        { unsigned v = 0u;
#pragma unroll
          for (int c = 0; c < 8; c++) v += hsub[c * 1024 + l];
          // inclusive prefix over the bins (thread = bin): hist[b] = candidates in bins 0 .. b
          int tot; const int ex = block_excl_scan<1024>((int)v, ws, &tot); hist[l] = (unsigned)ex + v; }
        __syncthreads();
        if (l == 0) tq[1] = clock64();
    }
    int nacc = 0;
    // one corner past the capacity is enough to know the result does not fit (it is counted, not stored)
    const int limit = (a.max_corners > 0) ? min(a.max_corners, a.cap + 1) : a.cap + 1;
    bool overflow = false;
    for (;;) {
        int count = total;                                   // the radix form: one "band" = every candidate, already in order
        if (SORTLDS) {
            // the next band: bins [lo, hi) with at most GFTT_BAND candidates, at least one bin (thread = bin, hist = inclusive prefix)
            {
                const int lo = s_hi;                              // (read by every thread before thread 0 rewrites it below the barrier)
                const unsigned before = lo > 0 ? hist[lo - 1] : 0u;
                const bool fits = l >= lo && hist[l] - before <= (unsigned)GFTT_BAND;
                const unsigned long long mk = __ballot(fits);     // bins are monotone: the fitting ones form a prefix of [lo, 1024)
                if ((l & 63) == 0) ws[l >> 6] = __popcll(mk);
                __syncthreads();
                if (l == 0) {
                    int nfit = 0;
                    for (int q = 0; q < NT / 64; q++) nfit += ws[q];
                    int hi = lo + nfit;
                    if (total == 0 || lo >= 1024) s_state = 1;
                    else if (nfit == 0) { hi = lo + 1; s_state = 2; }     // one bin of (nearly) equal values is larger than the band
                    s_lo = lo; s_hi = hi;
                }
            }
            __syncthreads();
            if (s_state) break;
            const unsigned lo = (unsigned)s_lo, hi = (unsigned)s_hi;
            const unsigned before = lo > 0 ? hist[lo - 1] : 0u;      // candidates in the bins above this band's
            if (l == 0) tq[2] -= clock64();
            nbands += l == 0 ? 1 : 0;
            // counting sort by bin (the bins ARE value order): a key goes to its bin's segment of the band, any place inside it …
            for (unsigned b = lo + (unsigned)l; b < hi; b += NT) hsub[b] = 0u;          // (hsub[0 .. 1024) serves as the bins' cursors now)
            __syncthreads();
            auto place = [&](unsigned k, int i) {
                const unsigned b = (k - s_kmin) >> shift;
                if (b >= lo && b < hi) {
                    const unsigned seg = (b > 0 ? hist[b - 1] : 0u) - before;
                    band[seg + atomicAdd(&hsub[b], 1u)] = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - a.valsA[i]);
                }
            };
#pragma unroll
            for (int j = 0; j < KPT; j++) if (l + j * NT < total) place(kreg[j], l + j * NT);
            for (int i = l + KPT * NT; i < total; i += NT) place(a.keysA[i], i);
            __syncthreads();
            count = (int)(hist[hi - 1] - before);
            {   // … and is then ranked among the few keys of its own bin (unique keys: the count of smaller ones is its place)
                unsigned long long mine = ~0ull; int place = 0;
                if (l < count) {
                    mine = band[l];
                    const unsigned b = ((unsigned)(mine >> 32) - s_kmin) >> shift;
                    const int s0 = (int)((b > 0 ? hist[b - 1] : 0u) - before), s1 = (int)(hist[b] - before);
                    int r = 0;
                    for (int j = s0; j < s1; j++) r += band[j] < mine ? 1 : 0;
                    place = s0 + r;
                }
                __syncthreads();
                if (l < count) band[place] = mine;
                __syncthreads();
            }
            if (l == 0) { const long long t = clock64(); tq[2] += t; tq[3] -= t; }
        }
        if (l < 64) {
            auto cand = [&](int i) -> unsigned { return SORTLDS ? 0xFFFFFFFFu - (unsigned)band[i] : vals[i]; };
            unsigned pnext = l < count ? cand(l) : 0u;       // the next batch's candidates are in flight while this one is resolved
            for (int base = 0; base < count && nacc < limit; base += 64) {
                const int i = base + l;
                bool alive = i < count;
                int x = 0, y = 0;
                const int p = (int)pnext;
                pnext = i + 64 < count ? cand(i + 64) : 0u;
                if (alive) { y = p / a.w; x = p - y * a.w; }
                const int xc = x / cell, yc = y / cell;
                if (alive && filter) {
                    if (grid) {
                        for (int dyc = -1; dyc <= 1 && alive; dyc++) {
                            const int cy = yc + dyc;
                            if (cy < 0 || cy >= gh) continue;
                            for (int dxc = -1; dxc <= 1 && alive; dxc++) {
                                const int cx = xc + dxc;
                                if (cx < 0 || cx >= gw) continue;
                                const int c = cy * gw + cx, n = min((int)gcnt[c], CELL_CAP);
                                for (int e = 0; e < n; e++) {
                                    const int k = gent[c * CELL_CAP + e];
                                    const float dx = (float)(x - ax[k]), dy = (float)(y - ay[k]);
                                    if ((double)(dx * dx + dy * dy) < md2) { alive = false; break; }
                                }
                            }
                        }
                    } else {
                        for (int k = 0; k < nacc; k++) {
                            const int dxc = acx[k] - xc, dyc = acy[k] - yc;
                            if (dxc < -1 || dxc > 1 || dyc < -1 || dyc > 1) continue;
                            const float dx = (float)(x - ax[k]), dy = (float)(y - ay[k]);
                            if ((double)(dx * dx + dy * dy) < md2) { alive = false; break; }
                        }
                    }
                }
                // accept rounds in registers: the strongest surviving lane is accepted, the lanes within min_dist of it drop out (a corner
                // that close always lies in a neighbouring cell, so the plain distance test equals the grid's) — no LDS inside the loop
                // (a table update per round was ~1 000 cycles: 68 us for 150 corners)
                uint64_t m = __ballot(alive), accm = 0ull;
                const int nacc0 = nacc;
                while (m && nacc < limit) {
                    const int first = __ffsll((long long)m) - 1;
                    const int fx = __builtin_amdgcn_readlane(x, first), fy = __builtin_amdgcn_readlane(y, first);      // `first` is wave-uniform: scalar reads
                    accm |= 1ull << first;
                    nacc++;
                    if (l == first) alive = false;
                    if (alive && filter) {
                        const float dx = (float)(x - fx), dy = (float)(y - fy);
                        if ((double)(dx * dx + dy * dy) < md2) alive = false;
                    }
                    m = __ballot(alive);
                    if (nacc >= ACC_MAX && filter) { overflow = true; break; }
                }
                // … then the batch's accepted corners enter the tables together (their order inside a cell does not matter)
                bool full = false;
                if ((accm >> l) & 1ull) {
                    const int k = nacc0 + __popcll(accm & ((1ull << l) - 1ull));
                    if (k < ACC_MAX) { ax[k] = (short)x; ay[k] = (short)y; acx[k] = (short)xc; acy[k] = (short)yc; }
                    if (k < a.cap) { a.out_xy[2 * k] = (float)x; a.out_xy[2 * k + 1] = (float)y; }
                    if (grid) {
                        const int c = yc * gw + xc;
                        const unsigned n = k < ACC_MAX ? atomicAdd(&gcnt[c], 1u) : (unsigned)CELL_CAP;
                        if (n < (unsigned)CELL_CAP) gent[c * CELL_CAP + n] = (unsigned short)k;
                        else full = true;
                    }
                }
                if (__ballot(full)) grid = false;                   // a fifth corner in one cell: from here on the walk over all accepted corners
                if (overflow) break;
                __threadfence_block(); __builtin_amdgcn_wave_barrier();      // one wavefront: its LDS operations complete in order
            }
            if (SORTLDS && l == 0) tq[3] += clock64();
            if (SORTLDS && l == 0 && (overflow || nacc >= limit)) s_state = overflow ? 2 : 1;
        }
        if (!SORTLDS) break;
        __syncthreads();
        if (s_state) break;
    }
    if (l == 0) *a.out_n = SORTLDS ? (s_state == 2 ? -2 : nacc) : (overflow ? -1 : nacc);
    if (SORTLDS && l == 0 && a.dbg) { a.dbg[0] = tq[1] - tq[0]; a.dbg[1] = tq[2]; a.dbg[2] = tq[3]; a.dbg[3] = clock64() - tq[0]; a.dbg[4] = nbands; a.dbg[5] = total; a.dbg[6] = nacc; }
}

int32_t tfail(int32_t code, const std::string& msg) { set_error(msg); return code; }

}  // namespace

}  // namespace lvi

using namespace lvi;

constexpr int CLAHE_MAX_TILES = 64;

struct lvi_tracker {
    lvi_tracker_params P;
    int device = 0;
    Ctx ctx; Profiler prof; Arena arena;
    Pyr pyr[2];                 // [cur, forw]
    int cur = 0, forw = 1;
    int w = 0, h = 0;
    bool have_forw = false, have_cur = false, have_lk = false, have_gftt = false, have_mask = false;
    uint8_t* d_stage = nullptr;
    float *d_cur_xy = nullptr, *d_forw_xy = nullptr, *d_err = nullptr; uint8_t* d_status = nullptr; int n_pts = 0;
    uint8_t* d_mask = nullptr; float* d_eig = nullptr; unsigned* d_maxord = nullptr; unsigned* d_maxPartial = nullptr; float* d_thr = nullptr;
    int *d_blockCnt = nullptr, *d_total = nullptr, *d_n = nullptr, *d_nbits = nullptr, *d_out_n = nullptr, *d_ncand = nullptr;
    float* d_gftt_xy = nullptr;
    SortPlan sort;
    int gftt_n = 0, gftt_ncand = 0;
    bool gftt_force_radix = false;     // LVI_GFTT_RADIX=1 at create: always the 12-launch radix sort + one-wavefront pick (tests: same corners)
    bool gftt_pending = false; int gftt_pending_max = 0;       // lvi_tracker_run_gftt_async enqueued, result not fetched yet
    float* d_centers = nullptr; float* h_centers[2] = {nullptr, nullptr}; hipEvent_t ev_centers[2] = {nullptr, nullptr}; int centers_slot = 0;
    float* d_frame_out = nullptr;      // [4 + 2 F + 2 F] one block: {n_new, n_cand, n_all, 0}, the new corners, the undistorted points of [kept ; new]
    float* h_frame_out = nullptr;      // pinned mirror
    float* d_all_xy = nullptr;         // [2 F] kept ++ new
    long long* d_dbg = nullptr;        // [8] phase stamps of gftt_sortpick (LVI_TDBG_SORTPICK_CYCLES)
    // f-2 / f-3
    uint8_t *d_eq = nullptr, *d_lut = nullptr; float *d_un_in = nullptr, *d_un_out = nullptr;
    bool equalize = false; double clahe_clip = 3.0; int clahe_tx = 8, clahe_ty = 8;
    // pinned staging for incoming frames: the caller's (pageable) image is copied here on the host and goes to the
    // device from pinned memory — measured: the runtime's own handling of a pageable 0.9 MB source costs up to 1 ms per
    // frame in a process with many GPU mappings, and the call had to wait for it before returning
    uint8_t* h_frame[2] = {nullptr, nullptr}; hipEvent_t ev_frame[2] = {nullptr, nullptr}; int frame_slot = 0;
    float* h_pts[2] = {nullptr, nullptr}; hipEvent_t ev_pts[2] = {nullptr, nullptr}; int pts_slot = 0;      // the same for cur_pts
    // Small inputs and results do not travel through copy commands (a 1 KB hipMemcpyAsync is a 4 us blit kernel on the stream, and the node
    // path had eleven of them per frame): kernels read the caller's points / circle centres in place from the pinned slots above and
    // write the LK results and the frame's result block into pinned host memory; the host waits once and reads.
    const float* cur_src = nullptr; int cur_slot = -1;       // points of the next run_lk (a pinned slot; its event is recorded behind the kernel that reads it)
    float* h_lk = nullptr;                                   // [2 F] forw_xy, [F] err, [F bytes] status
};

namespace {

template <class F>
int32_t tguard(lvi_tracker* t, F&& f)
{
    try {
        if (t) LVI_HIP(hipSetDevice(t->device));
        return f();
    } catch (const HipError& e) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", e.what, hipGetErrorString(e.e), e.file, e.line);
        return tfail(LVI_ERR_HIP, buf);
    }
}

template <class AR>
void tracker_layout(AR& ar, lvi_tracker& t)
{
    const int W = t.P.max_width, H = t.P.max_height;
    for (int s = 0; s < 2; s++) {
        int w = W, h = H;
        for (int l = 0; l <= t.P.lk_max_level; l++) {
            t.pyr[s].lv[l].px = ar.template alloc<uint8_t>((size_t)w * h);
            w = (w + 1) / 2; h = (h + 1) / 2;
        }
    }
    const int F = std::max(t.P.max_features, 64);
    t.d_cur_xy = ar.template alloc<float>(2 * (size_t)F); t.d_forw_xy = ar.template alloc<float>(2 * (size_t)F);
    t.d_err = ar.template alloc<float>(F); t.d_status = ar.template alloc<uint8_t>(F);
    t.d_mask = ar.template alloc<uint8_t>((size_t)W * H); t.d_eig = ar.template alloc<float>((size_t)W * H);
    t.d_maxord = ar.template alloc<unsigned>(1); t.d_thr = ar.template alloc<float>(1);
    t.d_maxPartial = ar.template alloc<unsigned>((size_t)div_up(W, 32) * div_up(H, 8));
    t.d_blockCnt = ar.template alloc<int>(div_up(W * H, CAND_TILE) + 1);
    t.d_total = ar.template alloc<int>(1); t.d_n = ar.template alloc<int>(1); t.d_nbits = ar.template alloc<int>(1);
    t.d_out_n = ar.template alloc<int>(1); t.d_ncand = ar.template alloc<int>(1);
    t.d_gftt_xy = ar.template alloc<float>(2 * (size_t)F);
    t.sort.allocate(ar, 1, W * H, RS_ITEMS_SMALL);
    t.d_stage = ar.template alloc<uint8_t>((size_t)W * H); t.d_eq = ar.template alloc<uint8_t>((size_t)W * H);
    t.d_lut = ar.template alloc<uint8_t>((size_t)CLAHE_MAX_TILES * CLAHE_MAX_TILES * 256);
    t.d_un_in = ar.template alloc<float>(2 * (size_t)F); t.d_un_out = ar.template alloc<float>(2 * (size_t)F);
    t.d_centers = ar.template alloc<float>(2 * (size_t)F); t.d_all_xy = ar.template alloc<float>(2 * (size_t)F);
    t.d_frame_out = ar.template alloc<float>(4 + 4 * (size_t)F);
    t.d_dbg = ar.template alloc<long long>(8);
}

// src (w x h, dense) → dst equalised; both device buffers of the handle
void run_clahe(lvi_tracker& t, const uint8_t* src, uint8_t* dst, int w, int h, double clip, int tilesX, int tilesY)
{
    int extW = w, extH = h;
    if (w % tilesX != 0 || h % tilesY != 0) { extW = w + (tilesX - (w % tilesX)); extH = h + (tilesY - (h % tilesY)); }   // copyMakeBorder quirk: both sides grow
    ClaheArgs a{};
    a.src = src; a.dst = dst; a.lut = t.d_lut; a.W = w; a.H = h; a.tilesX = tilesX; a.tilesY = tilesY;
    a.tw = extW / tilesX; a.th = extH / tilesY;
    const int area = a.tw * a.th;
    a.lutScale = static_cast<float>(255) / area;
    a.clipLimit = 0;
    if (clip > 0.0) a.clipLimit = std::max(static_cast<int>(clip * area / 256), 1);
    LVI_LAUNCH(t.ctx, "clahe_lut", (double)w * h, hipLaunchKernelGGL(clahe_lut_kernel, dim3(tilesX * tilesY), dim3(1024), 0, t.ctx.stream, a));
    LVI_LAUNCH(t.ctx, "clahe_interp", 2.0 * w * h, hipLaunchKernelGGL(clahe_interp_kernel, dim3(div_up(w, 64), div_up(h, 4)), dim3(256), 0, t.ctx.stream, a));
}

void build_pyramid(lvi_tracker& t, int slot)
{
    Pyr& p = t.pyr[slot];
    int w = t.w, h = t.h;
    p.lv[0].w = w; p.lv[0].h = h;
    p.top = t.P.lk_max_level;
    for (int level = 0; level <= t.P.lk_max_level; level++) {
        if (level != 0) {
            const int dw = (p.lv[level - 1].w + 1) / 2, dh = (p.lv[level - 1].h + 1) / 2;
            p.lv[level].w = dw; p.lv[level].h = dh;
            LVI_LAUNCH(t.ctx, "pyrdown", (double)p.lv[level - 1].w * p.lv[level - 1].h + (double)dw * dh,
                       hipLaunchKernelGGL(pyrdown_kernel, dim3(div_up(dw, 32), div_up(dh, 8)), dim3(256), 0, t.ctx.stream,
                                          p.lv[level - 1].px, p.lv[level - 1].w, p.lv[level - 1].h, p.lv[level].px, dw, dh));
        }
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= t.P.lk_win || h <= t.P.lk_win) { p.top = level; break; }      // buildOpticalFlowPyramid's early return
    }
}

}  // namespace

extern "C" {

void lvi_tracker_params_default(lvi_tracker_params* p)
{
    memset(p, 0, sizeof(*p));
    p->max_width = 1280; p->max_height = 720; p->max_cnt = 150; p->min_dist = 20.0;
    p->lk_win = 21; p->lk_max_level = 3; p->lk_max_iters = 30; p->lk_eps = 0.01; p->lk_min_eig_threshold = 1e-4f;
    p->gftt_quality = 0.01; p->max_features = 1024;
}

int32_t lvi_tracker_create(const lvi_tracker_params* p, int32_t device, lvi_tracker** out)
{
    if (!p || !out) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    if (p->lk_win < 3 || (p->lk_win & 1) == 0 || p->lk_win > LK_WIN_MAX || p->lk_max_level < 0 || p->lk_max_level >= MAX_LEVELS)
        return tfail(LVI_ERR_INVALID_ARG, "bad LK parameters (odd window <= 21, maxLevel 0..7)");
    if (p->max_width <= 0 || p->max_height <= 0 || p->max_features <= 0 || p->max_features > 4096) return tfail(LVI_ERR_INVALID_ARG, "bad capacities");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tfail(LVI_ERR_NO_DEVICE, "no HIP device: the HIP path has no CPU fallback");
    if (device < 0 || device >= ndev) return tfail(LVI_ERR_NO_DEVICE, "device index out of range");
    lvi_tracker* t = new lvi_tracker();
    t->P = *p; t->device = device;
    { const char* e = getenv("LVI_GFTT_RADIX"); t->gftt_force_radix = e && e[0] == '1'; }
    int32_t st = tguard(t, [&]() -> int32_t {
        LVI_HIP(hipStreamCreateWithFlags(&t->ctx.stream, hipStreamNonBlocking));
        t->ctx.prof = &t->prof;
        ArenaSizer sz; tracker_layout(sz, *t);
        t->arena.init(sz.used + (1 << 16));
        tracker_layout(t->arena, *t);
        LVI_HIP(hipMemsetAsync(t->arena.base, 0, t->arena.size, t->ctx.stream));
        for (int s = 0; s < 2; s++) {
            LVI_HIP(hipHostMalloc((void**)&t->h_frame[s], (size_t)t->P.max_width * t->P.max_height, hipHostMallocDefault));
            LVI_HIP(hipEventCreateWithFlags(&t->ev_frame[s], hipEventDisableTiming));
            LVI_HIP(hipEventRecord(t->ev_frame[s], t->ctx.stream));
            LVI_HIP(hipHostMalloc((void**)&t->h_centers[s], sizeof(float) * 2 * (size_t)std::max(t->P.max_features, 64), hipHostMallocDefault));
            LVI_HIP(hipEventCreateWithFlags(&t->ev_centers[s], hipEventDisableTiming));
            LVI_HIP(hipEventRecord(t->ev_centers[s], t->ctx.stream));
            LVI_HIP(hipHostMalloc((void**)&t->h_pts[s], sizeof(float) * 2 * (size_t)std::max(t->P.max_features, 64), hipHostMallocDefault));
            LVI_HIP(hipEventCreateWithFlags(&t->ev_pts[s], hipEventDisableTiming));
            LVI_HIP(hipEventRecord(t->ev_pts[s], t->ctx.stream));
        }
        LVI_HIP(hipHostMalloc((void**)&t->h_frame_out, sizeof(float) * (4 + 4 * (size_t)std::max(t->P.max_features, 64)), hipHostMallocDefault));
        LVI_HIP(hipHostMalloc((void**)&t->h_lk, sizeof(float) * 4 * (size_t)std::max(t->P.max_features, 64), hipHostMallocDefault));
        LVI_HIP(hipStreamSynchronize(t->ctx.stream));
        return LVI_OK;
    });
    if (st != LVI_OK) { lvi_tracker_destroy(t); return st; }
    *out = t;
    return LVI_OK;
}

void lvi_tracker_destroy(lvi_tracker* t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->ctx.stream) (void)hipStreamSynchronize(t->ctx.stream);
    t->prof.collect();
    t->arena.release();
    for (int s = 0; s < 2; s++) {
        if (t->h_frame[s]) (void)hipHostFree(t->h_frame[s]);
        if (t->ev_frame[s]) (void)hipEventDestroy(t->ev_frame[s]);
        if (t->h_pts[s]) (void)hipHostFree(t->h_pts[s]);
        if (t->h_centers[s]) (void)hipHostFree(t->h_centers[s]);
        if (t->ev_centers[s]) (void)hipEventDestroy(t->ev_centers[s]);
        if (t->ev_pts[s]) (void)hipEventDestroy(t->ev_pts[s]);
    }
    if (t->h_frame_out) (void)hipHostFree(t->h_frame_out);
    if (t->h_lk) (void)hipHostFree(t->h_lk);
    if (t->ctx.stream) (void)hipStreamDestroy(t->ctx.stream);
    delete t;
}

int32_t lvi_tracker_sync(lvi_tracker* t)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null handle");
    return tguard(t, [&]() -> int32_t { LVI_HIP(hipStreamSynchronize(t->ctx.stream)); return LVI_OK; });
}

int32_t lvi_tracker_push_image(lvi_tracker* t, const uint8_t* img, int32_t w, int32_t h, int32_t stride)
{
    if (!t || !img || w <= 0 || h <= 0 || stride < w) return tfail(LVI_ERR_INVALID_ARG, "bad image");
    if (w > t->P.max_width || h > t->P.max_height) return tfail(LVI_ERR_CAPACITY, "image exceeds capacity");
    return tguard(t, [&]() -> int32_t {
        if (t->have_forw && (w != t->w || h != t->h)) { t->have_forw = t->have_cur = false; }
        if (t->have_forw) { std::swap(t->cur, t->forw); t->have_cur = true; }     // cur_img = forw_img (:203)
        t->w = w; t->h = h;
        const int slot = (t->frame_slot ^= 1);
        LVI_HIP(hipEventSynchronize(t->ev_frame[slot]));                           // the copy that last used this staging buffer (two frames ago)
        for (int y = 0; y < h; y++) std::memcpy(t->h_frame[slot] + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
        uint8_t* dst0 = t->equalize ? t->d_stage : t->pyr[t->forw].lv[0].px;
        LVI_HIP(hipMemcpyAsync(dst0, t->h_frame[slot], (size_t)w * h, hipMemcpyHostToDevice, t->ctx.stream));
        LVI_HIP(hipEventRecord(t->ev_frame[slot], t->ctx.stream));
        if (t->equalize) run_clahe(*t, t->d_stage, t->pyr[t->forw].lv[0].px, w, h, t->clahe_clip, t->clahe_tx, t->clahe_ty);   // readImage's EQUALIZE branch (:86-90)
        build_pyramid(*t, t->forw);
        if (!t->have_forw) {                                                       // prev = cur = forw = img (:94-97)
            LVI_HIP(hipMemcpyAsync(t->pyr[t->cur].lv[0].px, t->pyr[t->forw].lv[0].px, (size_t)w * h, hipMemcpyDeviceToDevice, t->ctx.stream));
            build_pyramid(*t, t->cur);
            t->have_cur = true;
        }
        // no stream sync: the caller's buffer was consumed by the host copy above
        t->have_forw = true; t->have_lk = false; t->have_gftt = false;
        return LVI_OK;
    });
}

int32_t lvi_clahe(lvi_tracker* t, const uint8_t* img, int32_t w, int32_t h, int32_t stride, double clip_limit, int32_t tiles_x, int32_t tiles_y,
                  uint8_t* out, int32_t out_stride)
{
    if (!t || !img || !out || w <= 0 || h <= 0 || stride < w || out_stride < w) return tfail(LVI_ERR_INVALID_ARG, "bad image");
    if (tiles_x < 1 || tiles_y < 1 || tiles_x > CLAHE_MAX_TILES || tiles_y > CLAHE_MAX_TILES) return tfail(LVI_ERR_INVALID_ARG, "bad tile grid");
    if (w > t->P.max_width || h > t->P.max_height) return tfail(LVI_ERR_CAPACITY, "image exceeds capacity");
    return tguard(t, [&]() -> int32_t {
        LVI_HIP(hipMemcpy2DAsync(t->d_stage, (size_t)w, img, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, t->ctx.stream));
        run_clahe(*t, t->d_stage, t->d_eq, w, h, clip_limit, tiles_x, tiles_y);
        LVI_HIP(hipMemcpy2DAsync(out, (size_t)out_stride, t->d_eq, (size_t)w, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, t->ctx.stream));
        LVI_HIP(hipStreamSynchronize(t->ctx.stream));
        return LVI_OK;
    });
}

int32_t lvi_tracker_set_equalize(lvi_tracker* t, int32_t on, double clip_limit, int32_t tiles_x, int32_t tiles_y)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null handle");
    if (on && (tiles_x < 1 || tiles_y < 1 || tiles_x > CLAHE_MAX_TILES || tiles_y > CLAHE_MAX_TILES)) return tfail(LVI_ERR_INVALID_ARG, "bad tile grid");
    t->equalize = on != 0; t->clahe_clip = clip_limit; t->clahe_tx = tiles_x; t->clahe_ty = tiles_y;
    return LVI_OK;
}

int32_t lvi_undistort_points(lvi_tracker* t, const lvi_mei_params* cam, const float* xy, int32_t n, float* un_xy)
{
    if (!t || !cam || n < 0 || (n > 0 && (!xy || !un_xy))) return tfail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (n > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "too many points");
    if (n == 0) return LVI_OK;
    return tguard(t, [&]() -> int32_t {
        LVI_HIP(hipMemcpyAsync(t->d_un_in, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, t->ctx.stream));
        LVI_LAUNCH(t->ctx, "mei_undistort", 16.0 * n, hipLaunchKernelGGL(mei_undistort_kernel, dim3(div_up(n, 64)), dim3(64), 0, t->ctx.stream, *cam, t->d_un_in, n, t->d_un_out, (const int*)nullptr));
        LVI_HIP(hipMemcpyAsync(un_xy, t->d_un_out, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, t->ctx.stream));
        LVI_HIP(hipStreamSynchronize(t->ctx.stream));
        return LVI_OK;
    });
}

int32_t lvi_tracker_set_points(lvi_tracker* t, const float* cur_xy, int32_t n)
{
    if (!t || n < 0 || (n > 0 && !cur_xy)) return tfail(LVI_ERR_INVALID_ARG, "bad points");
    if (n > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "too many points");
    return tguard(t, [&]() -> int32_t {
        if (n) {
            const int slot = (t->pts_slot ^= 1);
            LVI_HIP(hipEventSynchronize(t->ev_pts[slot]));
            std::memcpy(t->h_pts[slot], cur_xy, sizeof(float) * 2 * (size_t)n);
            t->cur_src = t->h_pts[slot]; t->cur_slot = slot;               // lk_kernel reads them in place
        }
        t->n_pts = n; t->have_lk = false;
        return LVI_OK;
    });
}

int32_t lvi_tracker_run_lk(lvi_tracker* t)
{
    if (!t || !t->have_forw || !t->have_cur) return tfail(LVI_ERR_STATE, "no image pair");
    return tguard(t, [&]() -> int32_t {
        LkArgs a{};
        a.prev = t->pyr[t->cur]; a.next = t->pyr[t->forw];
        const size_t F = (size_t)std::max(t->P.max_features, 64);
        a.prev_xy = t->cur_src ? t->cur_src : t->d_cur_xy; a.next_xy = t->d_forw_xy; a.status = t->d_status; a.err = t->d_err;
        a.h_next_xy = t->h_lk; a.h_err = t->h_lk + 2 * F; a.h_status = reinterpret_cast<uint8_t*>(t->h_lk + 3 * F);
        a.n = t->n_pts; a.win = t->P.lk_win; a.max_level = std::min(a.prev.top, a.next.top);
        a.max_count = std::min(std::max(t->P.lk_max_iters, 0), 100);
        double eps = std::min(std::max(t->P.lk_eps, 0.), 10.);
        a.epsilon = eps * eps; a.min_eig = t->P.lk_min_eig_threshold;
        if (a.n > 0) {
            const double bytes = (double)a.n * (a.max_level + 1) * (24.0 * 24 + 22.0 * 22 * 4);
            LVI_LAUNCH(t->ctx, "lk_track", bytes, hipLaunchKernelGGL(lk_kernel, dim3(a.n), dim3(64), 0, t->ctx.stream, a));
            if (t->cur_slot >= 0) LVI_HIP(hipEventRecord(t->ev_pts[t->cur_slot], t->ctx.stream));      // the slot is free again once this kernel has read it
        }
        t->have_lk = true;
        return LVI_OK;
    });
}

int32_t lvi_tracker_get_lk(lvi_tracker* t, float* forw_xy, uint8_t* status, float* err, int32_t capacity, int32_t* n)
{
    if (!t || !n) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    if (!t->have_lk) return tfail(LVI_ERR_STATE, "LK not run");
    *n = t->n_pts;
    if (capacity < *n) return tfail(LVI_ERR_CAPACITY, "capacity too small");
    return tguard(t, [&]() -> int32_t {
        const int m = *n;
        const size_t F = (size_t)std::max(t->P.max_features, 64);
        LVI_HIP(hipStreamSynchronize(t->ctx.stream));                      // lk_kernel wrote the results into pinned host memory
        if (m && forw_xy) std::memcpy(forw_xy, t->h_lk, sizeof(float) * 2 * (size_t)m);
        if (m && err) std::memcpy(err, t->h_lk + 2 * F, sizeof(float) * (size_t)m);
        if (m && status) std::memcpy(status, t->h_lk + 3 * F, (size_t)m);
        return LVI_OK;
    });
}

int32_t lvi_tracker_set_mask(lvi_tracker* t, const uint8_t* mask, int32_t w, int32_t h, int32_t stride)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    if (!mask) { t->have_mask = false; return LVI_OK; }
    if (!t->have_forw || w != t->w || h != t->h || stride < w) return tfail(LVI_ERR_INVALID_ARG, "mask size mismatch");
    return tguard(t, [&]() -> int32_t {
        LVI_HIP(hipMemcpy2DAsync(t->d_mask, (size_t)w, mask, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, t->ctx.stream));
        LVI_HIP(hipStreamSynchronize(t->ctx.stream));
        t->have_mask = true;
        return LVI_OK;
    });
}

namespace {
// goodFeaturesToTrack on forw, enqueued only.  lds_form: the candidates are sorted and picked by ONE workgroup in LDS (a frame with
// more than GFTT_LDS_MAX candidates reports -2 in d_out_n and is redone in the radix form by whoever fetches the result).
void enqueue_gftt(lvi_tracker* t, int32_t max_corners, bool lds_form)
{
    const int w = t->w, h = t->h, npx = w * h;
    GfttArgs a{};
    a.img = t->pyr[t->forw].lv[0].px; a.mask = t->have_mask ? t->d_mask : nullptr; a.w = w; a.h = h;
    a.eig = t->d_eig; a.maxord = t->d_maxord; a.maxPartial = t->d_maxPartial; a.thr = t->d_thr; a.blockCnt = t->d_blockCnt; a.total = t->d_total;
    a.keysA = t->sort.keysA; a.valsA = t->sort.valsA; a.d_n = t->d_n; a.d_nbits = t->d_nbits; a.quality = t->P.gftt_quality;
    const int nblk = div_up(npx, CAND_TILE);
    LVI_LAUNCH(t->ctx, "gftt_mineig", 2.0 * npx + 4.0 * npx, hipLaunchKernelGGL(mineig_kernel, dim3(div_up(w, 32), div_up(h, 8)), dim3(256), 0, t->ctx.stream, a));
    LVI_LAUNCH(t->ctx, "gftt_count", 5.0 * npx, hipLaunchKernelGGL(gftt_count_kernel, dim3(nblk), dim3(256), 0, t->ctx.stream, a, div_up(w, 32) * div_up(h, 8)));
    LVI_LAUNCH(t->ctx, "gftt_emit", 5.0 * npx, hipLaunchKernelGGL(gftt_emit_kernel, dim3(nblk), dim3(256), 0, t->ctx.stream, a, nblk));
    PickArgs p{};
    p.keysA = t->sort.keysA; p.valsA = t->sort.valsA; p.valsB = t->sort.valsB; p.d_nbits = t->d_nbits; p.total = t->d_total;
    p.w = w; p.h = h; p.max_corners = max_corners; p.cap = t->P.max_features; p.min_dist = t->P.min_dist;
    p.out_xy = t->d_gftt_xy; p.out_n = t->d_out_n; p.ncand = t->d_ncand; p.dbg = t->d_dbg;
    if (lds_form) {
        LVI_LAUNCH(t->ctx, "gftt_sortpick", 0, hipLaunchKernelGGL(gftt_pick_kernel<true>, dim3(1), dim3(1024), 0, t->ctx.stream, p));
    } else {
        radix_sort_pairs(t->ctx, t->sort, t->d_n, t->d_nbits, 4, "gftt", 0.02 * npx);
        LVI_LAUNCH(t->ctx, "gftt_pick", 0, hipLaunchKernelGGL(gftt_pick_kernel<false>, dim3(1), dim3(64), 0, t->ctx.stream, p));
    }
}
}  // namespace

int32_t lvi_tracker_run_gftt(lvi_tracker* t, int32_t max_corners)
{
    if (!t || !t->have_forw) return tfail(LVI_ERR_STATE, "no image");
    return tguard(t, [&]() -> int32_t {
        int res[2] = {0, 0};
        for (int attempt = 0; attempt < 2; attempt++) {
            enqueue_gftt(t, max_corners, attempt == 0 && !t->gftt_force_radix);
            LVI_HIP(hipMemcpyAsync(&res[0], t->d_out_n, sizeof(int), hipMemcpyDeviceToHost, t->ctx.stream));
            LVI_HIP(hipMemcpyAsync(&res[1], t->d_ncand, sizeof(int), hipMemcpyDeviceToHost, t->ctx.stream));
            LVI_HIP(hipStreamSynchronize(t->ctx.stream));
            if (res[0] != -2) break;                        // -2: more candidates than the LDS form holds -> once more, radix form
        }
        t->gftt_pending = false;
        if (res[0] < 0) return tfail(LVI_ERR_CAPACITY, "more corners than the pick kernel's accepted-list capacity");
        if (res[0] > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "more corners than max_features");
        t->gftt_n = res[0]; t->gftt_ncand = res[1]; t->have_gftt = true;
        return LVI_OK;
    });
}

int32_t lvi_tracker_set_mask_circles(lvi_tracker* t, const float* centers_xy, int32_t n, int32_t radius)
{
    if (!t || n < 0 || (n > 0 && !centers_xy) || radius < 0 || radius > 120) return tfail(LVI_ERR_INVALID_ARG, "bad circle list");
    if (!t->have_forw) return tfail(LVI_ERR_STATE, "no image");
    if (n > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "too many circles");
    return tguard(t, [&]() -> int32_t {
        CircleArgs a{t->d_mask, t->w, t->h, t->d_centers, n, radius};
        int slot = -1;
        if (n) {
            slot = (t->centers_slot ^= 1);
            LVI_HIP(hipEventSynchronize(t->ev_centers[slot]));
            std::memcpy(t->h_centers[slot], centers_xy, sizeof(float) * 2 * (size_t)n);
            a.centers = t->h_centers[slot];                                 // read in place (pinned host memory)
        }
        LVI_LAUNCH(t->ctx, "mask_fill", (double)t->w * t->h, hipLaunchKernelGGL(mask_fill_kernel, dim3(div_up(div_up(t->w * t->h, 16), 256)), dim3(256), 0, t->ctx.stream, a));
        if (n) {
            LVI_LAUNCH(t->ctx, "mask_circles", 0, hipLaunchKernelGGL(mask_circles_kernel, dim3(n), dim3(64), 0, t->ctx.stream, a));
            LVI_HIP(hipEventRecord(t->ev_centers[slot], t->ctx.stream));
        }
        t->have_mask = true;
        return LVI_OK;
    });
}

int32_t lvi_tracker_run_gftt_async(lvi_tracker* t, int32_t max_corners)
{
    if (!t || !t->have_forw) return tfail(LVI_ERR_STATE, "no image");
    return tguard(t, [&]() -> int32_t {
        enqueue_gftt(t, max_corners, !t->gftt_force_radix);
        t->gftt_pending = true; t->gftt_pending_max = max_corners; t->have_gftt = false;
        return LVI_OK;
    });
}

int32_t lvi_tracker_finish_frame(lvi_tracker* t, const lvi_mei_params* cam, const float* kept_xy, int32_t n_kept,
                                 float* new_xy, int32_t new_capacity, int32_t* n_new, float* un_xy)
{
    if (!t || n_kept < 0 || (n_kept > 0 && !kept_xy) || !n_new) return tfail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (n_kept > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "too many points");
    return tguard(t, [&]() -> int32_t {
        const int F = t->P.max_features;
        for (int attempt = 0; attempt < 2; attempt++) {
            const bool with_gftt = t->gftt_pending;
            const float* kept_src = t->d_un_in;
            int slot = -1;
            if (n_kept) {
                slot = (t->pts_slot ^= 1);
                LVI_HIP(hipEventSynchronize(t->ev_pts[slot]));
                std::memcpy(t->h_pts[slot], kept_xy, sizeof(float) * 2 * (size_t)n_kept);
                kept_src = t->h_pts[slot];                                  // read in place
            }
            int* d_hdr = reinterpret_cast<int*>(t->d_frame_out);
            int* h_hdr = reinterpret_cast<int*>(t->h_frame_out);
            LVI_LAUNCH(t->ctx, "frame_concat", 0, hipLaunchKernelGGL(frame_concat_kernel, dim3(1), dim3(64), 0, t->ctx.stream, kept_src, n_kept, t->d_gftt_xy, t->d_out_n, t->d_ncand,
                                                                     with_gftt ? 1 : 0, F, t->d_all_xy, d_hdr, h_hdr, t->h_frame_out + 4));
            if (slot >= 0) LVI_HIP(hipEventRecord(t->ev_pts[slot], t->ctx.stream));
            if (cam) LVI_LAUNCH(t->ctx, "mei_undistort", 16.0 * F, hipLaunchKernelGGL(mei_undistort_kernel, dim3(div_up(F, 64)), dim3(64), 0, t->ctx.stream, *cam, t->d_all_xy, F,
                                                                                       t->h_frame_out + 4 + 2 * (size_t)F, (const int*)(d_hdr + 2)));
            LVI_HIP(hipStreamSynchronize(t->ctx.stream));                   // the ONE wait of the frame end: both kernels wrote into pinned host memory
            const int* hdr = reinterpret_cast<const int*>(t->h_frame_out);
            if (with_gftt && hdr[0] == -2 && attempt == 0) { enqueue_gftt(t, t->gftt_pending_max, false); continue; }   // beyond the LDS form: redo in the radix form
            t->gftt_pending = false;
            const int nn = with_gftt ? hdr[0] : 0;
            if (nn < 0) return tfail(LVI_ERR_CAPACITY, "more corners than the pick kernel's accepted-list capacity");
            if (nn > F || n_kept + nn > F) return tfail(LVI_ERR_CAPACITY, "more corners than max_features");
            if (nn > new_capacity) return tfail(LVI_ERR_CAPACITY, "capacity too small");
            *n_new = nn;
            if (with_gftt) { t->gftt_n = nn; t->gftt_ncand = hdr[1]; t->have_gftt = true; }
            if (nn && new_xy) std::memcpy(new_xy, t->h_frame_out + 4, sizeof(float) * 2 * (size_t)nn);
            if (cam && un_xy) std::memcpy(un_xy, t->h_frame_out + 4 + 2 * (size_t)F, sizeof(float) * 2 * (size_t)(n_kept + nn));
            return LVI_OK;
        }
        return LVI_OK;
    });
}

int32_t lvi_tracker_get_gftt(lvi_tracker* t, float* xy, int32_t capacity, int32_t* n)
{
    if (!t || !n) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    if (!t->have_gftt) return tfail(LVI_ERR_STATE, "GFTT not run");
    *n = t->gftt_n;
    if (capacity < *n) return tfail(LVI_ERR_CAPACITY, "capacity too small");
    return tguard(t, [&]() -> int32_t {
        if (*n && xy) LVI_HIP(hipMemcpyAsync(xy, t->d_gftt_xy, sizeof(float) * 2 * *n, hipMemcpyDeviceToHost, t->ctx.stream));
        LVI_HIP(hipStreamSynchronize(t->ctx.stream));
        return LVI_OK;
    });
}

int32_t lvi_lk_track(lvi_tracker* t, const uint8_t* prev, const uint8_t* next, int32_t w, int32_t h, int32_t stride,
                     const float* prev_xy, int32_t n, float* next_xy, uint8_t* status, float* err)
{
    if (!t || !prev || !next || n < 0) return tfail(LVI_ERR_INVALID_ARG, "bad arguments");
    t->have_forw = false; t->have_cur = false;
    int32_t st = lvi_tracker_push_image(t, prev, w, h, stride); if (st) return st;
    st = lvi_tracker_push_image(t, next, w, h, stride); if (st) return st;
    st = lvi_tracker_set_points(t, prev_xy, n); if (st) return st;
    st = lvi_tracker_run_lk(t); if (st) return st;
    int32_t m = 0;
    return lvi_tracker_get_lk(t, next_xy, status, err, n, &m);
}

int32_t lvi_good_features(lvi_tracker* t, const uint8_t* img, const uint8_t* mask, int32_t w, int32_t h, int32_t stride,
                          int32_t max_corners, double quality, double min_dist, float* xy, int32_t xy_capacity, int32_t* n_out)
{
    if (!t || !img || !n_out) return tfail(LVI_ERR_INVALID_ARG, "bad arguments");
    t->have_forw = false; t->have_cur = false;
    int32_t st = lvi_tracker_push_image(t, img, w, h, stride); if (st) return st;
    st = lvi_tracker_set_mask(t, mask, w, h, stride); if (st) return st;
    const double q0 = t->P.gftt_quality, d0 = t->P.min_dist;
    t->P.gftt_quality = quality; t->P.min_dist = min_dist;
    st = lvi_tracker_run_gftt(t, max_corners);
    t->P.gftt_quality = q0; t->P.min_dist = d0;
    if (st) return st;
    return lvi_tracker_get_gftt(t, xy, xy_capacity, n_out);
}

int32_t lvi_tracker_debug_get(lvi_tracker* t, int32_t what, void* dst, int64_t cap, int64_t* n_bytes)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null handle");
    return tguard(t, [&]() -> int32_t {
        const void* src = nullptr; int64_t bytes = 0; int32_t host_val = 0; bool host = false;
        switch (what) {
            case LVI_TDBG_PYRAMID_L1: case LVI_TDBG_PYRAMID_L2: case LVI_TDBG_PYRAMID_L3: {
                const int l = what - LVI_TDBG_PYRAMID_L1 + 1;
                if (!t->have_forw || l > t->pyr[t->forw].top) return tfail(LVI_ERR_STATE, "level not built");
                src = t->pyr[t->forw].lv[l].px; bytes = (int64_t)t->pyr[t->forw].lv[l].w * t->pyr[t->forw].lv[l].h;
                break;
            }
            case LVI_TDBG_MINEIG:
                if (!t->have_gftt) return tfail(LVI_ERR_STATE, "GFTT not run");
                src = t->d_eig; bytes = (int64_t)t->w * t->h * 4;
                break;
            case LVI_TDBG_GFTT_NCAND:
                if (!t->have_gftt) return tfail(LVI_ERR_STATE, "GFTT not run");
                host = true; host_val = t->gftt_ncand; bytes = 4;
                break;
            case 7:                                  // phase stamps of gftt_sortpick (diagnostics, HIP only)
                src = t->d_dbg; bytes = 64;
                break;
            case LVI_TDBG_MASK:
                if (!t->have_mask || !t->have_forw) return tfail(LVI_ERR_STATE, "no mask");
                src = t->d_mask; bytes = (int64_t)t->w * t->h;
                break;
            default: return tfail(LVI_ERR_INVALID_ARG, "unknown debug item");
        }
        if (n_bytes) *n_bytes = bytes;
        if (!dst) return LVI_OK;
        if (cap < bytes) return tfail(LVI_ERR_CAPACITY, "debug buffer too small");
        if (host) { memcpy(dst, &host_val, 4); return LVI_OK; }
        LVI_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, t->ctx.stream));
        LVI_HIP(hipStreamSynchronize(t->ctx.stream));
        return LVI_OK;
    });
}

int32_t lvi_tracker_prof_enable(lvi_tracker* t, int32_t on)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null handle");
    return tguard(t, [&]() -> int32_t { LVI_HIP(hipStreamSynchronize(t->ctx.stream)); t->prof.collect(); t->prof.on = on != 0; return LVI_OK; });
}
int32_t lvi_tracker_prof_reset(lvi_tracker* t)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null handle");
    return tguard(t, [&]() -> int32_t { LVI_HIP(hipStreamSynchronize(t->ctx.stream)); t->prof.reset(); return LVI_OK; });
}
int32_t lvi_tracker_prof_read(lvi_tracker* t, lvi_kernel_stat* stats, int32_t capacity, int32_t* n)
{
    if (!t || !n) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    return tguard(t, [&]() -> int32_t {
        Profiler& p = t->prof;
        LVI_HIP(hipStreamSynchronize(t->ctx.stream)); p.collect();
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
