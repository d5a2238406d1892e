// TEST INFRASTRUCTURE — CPU oracle of the feature_tracker hot path.
// Same C-ABI as the product (include/lvi_hotpath.h); only tests/, smoke() and
// bench.py's cpu_baseline leg may load it.
//
// PARITY UNPINNED.  The arithmetic lives in OpenCV (cv::calcOpticalFlowPyrLK,
// cv::goodFeaturesToTrack), which is neither vendored in the reference tree nor installed
// here; the reference only requires `find_package(OpenCV 4)` and has no tests.  This file
// restates the published OpenCV 4.x algorithms (video/src/lkpyramid.cpp,
// imgproc/src/{pyramids,corner,featureselect}.cpp) for the exact call sites
//   feature_tracker/src/feature_tracker.cpp:113   calcOpticalFlowPyrLK(cur,forw,pts,…,Size(21,21),3)
//   feature_tracker/src/feature_tracker.cpp:166   goodFeaturesToTrack(forw,n_pts,N,0.01,MIN_DIST,mask)
// Choices where OpenCV's result depends on the build:
//   * LK accumulators (A11,A12,A22,b1,b2): OpenCV's scalar code sums them in its `acctype` — float in
//     the build SURVEY App. A.6 describes — and its SIMD paths sum the same products in four lanes:
//     the float result depends on the build's summation order.  Two forms are restated:
//       mode 0 (default)  exact integer sums (int64), converted to float once: the value every float
//                         order approximates, order-independent, and therefore what a parallel
//                         implementation can reproduce bit for bit;
//       mode 1            SURVEY App. A.6 literally: float accumulators, one scalar addition per
//                         pixel in row-major order, then x 2^-20.
//     lvo_set_lk_accumulators(mode) switches (process-wide, test infrastructure only).  The measured
//     difference between the two (status flips at the minEig test, positions) is reported by
//     tests/test_gpu_tracker.py::test_lk_float_accumulator_variant_report and DESIGN.md 2.
//   * Sobel / min-eigenvalue arithmetic: scalar (non-SIMD, non-fused) operation order.
//   * single-threaded scalar code (real OpenCV uses SIMD and a thread pool — say so next to
//     any speed-up quoted against this baseline).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../include/lvi_hotpath.h"

void lvo_set_error(const char* msg);   // lvo_lidar.cpp

static int g_lk_acc_mode = 0;                   // 0: exact integer sums; 1: float sums, scalar row-major order
extern "C" void lvo_set_lk_accumulators(int mode) { g_lk_acc_mode = mode ? 1 : 0; }
extern "C" int lvo_get_lk_accumulators(void) { return g_lk_acc_mode; }

namespace {

int32_t tfail(int32_t code, const char* msg) { lvo_set_error(msg); return code; }

inline int cvFloor(float v) { int i = (int)v; return i - (i > v); }
inline int cvRound(float v) { return (int)std::lrintf(v); }          // round-half-even, as SSE/NEON cvRound
inline int cvRoundD(double v) { return (int)std::lrint(v); }
inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

struct Image8 {
    int w = 0, h = 0;
    std::vector<uint8_t> px;
    uint8_t at(int x, int y) const { return px[(size_t)y * w + x]; }
    // pyramid levels are stored with a REFLECT_101 border of winSize in OpenCV; reading through
    // reflect101 is the same thing
    uint8_t atR(int x, int y) const { return px[(size_t)reflect101(y, h) * w + reflect101(x, w)]; }
};

// cv::pyrDown (pyramids.cpp pyrDown_<FixPtCast<uchar,8>>): 5x5 [1 4 6 4 1]⊗[1 4 6 4 1], REFLECT_101,
// dst = (sum + 128) >> 8, size ((w+1)/2, (h+1)/2)
void pyrDown(const Image8& s, Image8& d)
{
    d.w = (s.w + 1) / 2; d.h = (s.h + 1) / 2;
    d.px.resize((size_t)d.w * d.h);
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < d.h; y++)
        for (int x = 0; x < d.w; x++) {
            int sum = 0;
            for (int j = 0; j < 5; j++) {
                int sy = reflect101(2 * y + j - 2, s.h);
                int rs = 0;
                for (int i = 0; i < 5; i++) rs += k[i] * s.px[(size_t)sy * s.w + reflect101(2 * x + i - 2, s.w)];
                sum += k[j] * rs;
            }
            d.px[(size_t)y * d.w + x] = (uint8_t)((sum + 128) >> 8);
        }
}

// buildOpticalFlowPyramid(img, pyr, winSize, maxLevel, withDerivatives=false): returns the top level
int buildPyramid(const Image8& img, std::vector<Image8>& pyr, int win, int maxLevel)
{
    pyr.clear(); pyr.push_back(img);
    int w = img.w, h = img.h;
    for (int level = 0; level <= maxLevel; ++level) {
        if (level != 0) { Image8 d; pyrDown(pyr[level - 1], d); pyr.push_back(std::move(d)); }
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win || h <= win) return level;
    }
    return maxLevel;
}

// calcScharrDeriv (lkpyramid.cpp): int16 (dx,dy) per pixel, REFLECT_101 at the image edge
struct Deriv16 {
    int w = 0, h = 0;
    std::vector<int16_t> d;     // 2 per pixel
    // the derivative image is padded with a CONSTANT 0 border of winSize
    int16_t at(int x, int y, int c) const { return (x < 0 || y < 0 || x >= w || y >= h) ? (int16_t)0 : d[((size_t)y * w + x) * 2 + c]; }
};
void scharr(const Image8& s, Deriv16& o)
{
    o.w = s.w; o.h = s.h; o.d.resize((size_t)s.w * s.h * 2);
    std::vector<int> t0(s.w + 2), t1(s.w + 2);
    for (int y = 0; y < s.h; y++) {
        int y0 = y > 0 ? y - 1 : (s.h > 1 ? 1 : 0);
        int y2 = y < s.h - 1 ? y + 1 : (s.h > 1 ? s.h - 2 : 0);
        for (int x = 0; x < s.w; x++) {
            int a = s.at(x, y0), b = s.at(x, y), c = s.at(x, y2);
            t0[x + 1] = (a + c) * 3 + b * 10;
            t1[x + 1] = c - a;
        }
        int x0 = s.w > 1 ? 1 : 0, x1 = s.w > 1 ? s.w - 2 : 0;
        t0[0] = t0[x0 + 1]; t0[s.w + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[s.w + 1] = t1[x1 + 1];
        for (int x = 0; x < s.w; x++) {
            o.d[((size_t)y * s.w + x) * 2 + 0] = (int16_t)(t0[x + 2] - t0[x]);
            o.d[((size_t)y * s.w + x) * 2 + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
}

#define LVO_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

// cv::detail::LKTrackerInvoker::operator() for one level, all points
void lkLevel(const Image8& I, const Deriv16& dI, const Image8& J, const float* prevPts, float* nextPts, uint8_t* status, float* err,
             int npoints, int win, int level, int maxLevel, int maxCount, double epsilon, float minEigThreshold)
{
    const float halfWin = (win - 1) * 0.5f;
    std::vector<int16_t> IWin((size_t)win * win), dIWin((size_t)win * win * 2);
    for (int p = 0; p < npoints; p++) {
        float prevx = prevPts[2 * p] * (float)(1. / (1 << level));
        float prevy = prevPts[2 * p + 1] * (float)(1. / (1 << level));
        float nextx, nexty;
        if (level == maxLevel) { nextx = prevx; nexty = prevy; }
        else { nextx = nextPts[2 * p] * 2.f; nexty = nextPts[2 * p + 1] * 2.f; }
        nextPts[2 * p] = nextx; nextPts[2 * p + 1] = nexty;

        prevx -= halfWin; prevy -= halfWin;
        int ipx = cvFloor(prevx), ipy = cvFloor(prevy);
        if (ipx < -win || ipx >= dI.w || ipy < -win || ipy >= dI.h) {
            if (level == 0) { status[p] = 0; err[p] = 0; }
            continue;
        }
        float a = prevx - ipx, b = prevy - ipy;
        const int W_BITS = 14, W_BITS1 = 14;
        const float FLT_SCALE = 1.f / (1 << 20);
        int iw00 = cvRound((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = cvRound(a * (1.f - b) * (1 << W_BITS));
        int iw10 = cvRound((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t iA11 = 0, iA12 = 0, iA22 = 0;
        float fA11 = 0.f, fA12 = 0.f, fA22 = 0.f;
        for (int y = 0; y < win; y++)
            for (int x = 0; x < win; x++) {
                int X = ipx + x, Y = ipy + y;
                int ival = LVO_DESCALE(I.atR(X, Y) * iw00 + I.atR(X + 1, Y) * iw01 + I.atR(X, Y + 1) * iw10 + I.atR(X + 1, Y + 1) * iw11, W_BITS1 - 5);
                int ixval = LVO_DESCALE(dI.at(X, Y, 0) * iw00 + dI.at(X + 1, Y, 0) * iw01 + dI.at(X, Y + 1, 0) * iw10 + dI.at(X + 1, Y + 1, 0) * iw11, W_BITS1);
                int iyval = LVO_DESCALE(dI.at(X, Y, 1) * iw00 + dI.at(X + 1, Y, 1) * iw01 + dI.at(X, Y + 1, 1) * iw10 + dI.at(X + 1, Y + 1, 1) * iw11, W_BITS1);
                IWin[(size_t)y * win + x] = (int16_t)ival;
                dIWin[((size_t)y * win + x) * 2] = (int16_t)ixval;
                dIWin[((size_t)y * win + x) * 2 + 1] = (int16_t)iyval;
                iA11 += (int)(ixval * ixval);
                iA12 += (int)(ixval * iyval);
                iA22 += (int)(iyval * iyval);
                fA11 += (float)(ixval * ixval);
                fA12 += (float)(ixval * iyval);
                fA22 += (float)(iyval * iyval);
            }
        float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
        if (g_lk_acc_mode == 1) { A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE; }
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
        if (minEig < minEigThreshold || D < FLT_EPSILON) {
            if (level == 0) status[p] = 0;
            continue;
        }
        D = 1.f / D;
        nextx -= halfWin; nexty -= halfWin;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < maxCount; j++) {
            int inx = cvFloor(nextx), iny = cvFloor(nexty);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                if (level == 0) status[p] = 0;
                break;
            }
            a = nextx - inx; b = nexty - iny;
            iw00 = cvRound((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cvRound(a * (1.f - b) * (1 << W_BITS));
            iw10 = cvRound((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t ib1 = 0, ib2 = 0;
            float fb1 = 0.f, fb2 = 0.f;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    int X = inx + x, Y = iny + y;
                    int diff = LVO_DESCALE(J.atR(X, Y) * iw00 + J.atR(X + 1, Y) * iw01 + J.atR(X, Y + 1) * iw10 + J.atR(X + 1, Y + 1) * iw11, W_BITS1 - 5)
                             - IWin[(size_t)y * win + x];
                    ib1 += (int)(diff * dIWin[((size_t)y * win + x) * 2]);
                    ib2 += (int)(diff * dIWin[((size_t)y * win + x) * 2 + 1]);
                    fb1 += (float)(diff * dIWin[((size_t)y * win + x) * 2]);
                    fb2 += (float)(diff * dIWin[((size_t)y * win + x) * 2 + 1]);
                }
            float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
            if (g_lk_acc_mode == 1) { b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE; }
            float dx = (float)((A12 * b2 - A22 * b1) * D);
            float dy = (float)((A12 * b1 - A11 * b2) * D);
            nextx += dx; nexty += dy;
            nextPts[2 * p] = nextx + halfWin; nextPts[2 * p + 1] = nexty + halfWin;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;      // Point2f::ddot is a double dot product
            if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
                nextPts[2 * p] -= dx * 0.5f; nextPts[2 * p + 1] -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status[p] && level == 0) {
            float nx = nextPts[2 * p] - halfWin, ny = nextPts[2 * p + 1] - halfWin;
            int inx = cvFloor(nx), iny = cvFloor(ny);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { status[p] = 0; continue; }
            float aa = nx - inx, bb = ny - iny;
            iw00 = cvRound((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cvRound(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cvRound((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float errval = 0.f;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    int X = inx + x, Y = iny + y;
                    int diff = LVO_DESCALE(J.atR(X, Y) * iw00 + J.atR(X + 1, Y) * iw01 + J.atR(X, Y + 1) * iw10 + J.atR(X + 1, Y + 1) * iw11, W_BITS1 - 5)
                             - IWin[(size_t)y * win + x];
                    errval += std::abs((float)diff);
                }
            err[p] = errval * 1.f / (32 * win * win);
        }
    }
}

// cv::cornerMinEigenVal(img, eig, blockSize=3, ksize=3, BORDER_DEFAULT) for CV_8U input
void cornerMinEigenVal(const Image8& s, std::vector<float>& eig)
{
    const int w = s.w, h = s.h;
    const double scale = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float k1 = (float)(1.0 * scale), k0 = (float)(2.0 * scale);
    std::vector<float> Dx((size_t)w * h), Dy((size_t)w * h);
    // Sobel dx: row filter [-1 0 1] (exact), column filter [1 2 1]*scale: (S0+S2)*k1 + S1*k0
    // Sobel dy: row filter [1 2 1]*scale: S[0]*k0 + (S[-1]+S[1])*k1, column filter [-1 0 1]
    std::vector<float> rowDiff((size_t)w * h), rowSmooth((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float l = (float)s.at(reflect101(x - 1, w), y), c = (float)s.at(x, y), r = (float)s.at(reflect101(x + 1, w), y);
            rowDiff[(size_t)y * w + x] = r - l;
            float t0 = c * k0; float t1 = (l + r) * k1;
            rowSmooth[(size_t)y * w + x] = t0 + t1;
        }
    for (int y = 0; y < h; y++) {
        int y0 = reflect101(y - 1, h), y2 = reflect101(y + 1, h);
        for (int x = 0; x < w; x++) {
            float S0 = rowDiff[(size_t)y0 * w + x], S1 = rowDiff[(size_t)y * w + x], S2 = rowDiff[(size_t)y2 * w + x];
            float u = (S0 + S2) * k1; float v = S1 * k0;
            Dx[(size_t)y * w + x] = u + v;
            Dy[(size_t)y * w + x] = rowSmooth[(size_t)y2 * w + x] - rowSmooth[(size_t)y0 * w + x];
        }
    }
    // cov = (dx*dx, dx*dy, dy*dy); boxFilter 3x3 un-normalised (f32 in, f64 accumulate, f32 out), REFLECT_101
    eig.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int j = -1; j <= 1; j++) {
                int yy = reflect101(y + j, h);
                for (int i = -1; i <= 1; i++) {
                    int xx = reflect101(x + i, w);
                    float dx = Dx[(size_t)yy * w + xx], dy = Dy[(size_t)yy * w + xx];
                    float xx2 = dx * dx, xy = dx * dy, yy2 = dy * dy;
                    s0 += xx2; s1 += xy; s2 += yy2;
                }
            }
            float a = (float)s0 * 0.5f, b = (float)s1, c = (float)s2 * 0.5f;
            float t = (a - c) * (a - c); float u = b * b;
            eig[(size_t)y * w + x] = (float)((a + c) - std::sqrt(t + u));
        }
}

struct GfttCand { float val; int addr; };

// cv::goodFeaturesToTrack (featureselect.cpp), Harris off
int goodFeatures(const Image8& img, const uint8_t* mask, int mstride, int maxCorners, double quality, double minDistance,
                 std::vector<float>& eig, float* xy, int xy_cap, int* ncand_out)
{
    const int w = img.w, h = img.h;
    cornerMinEigenVal(img, eig);
    double maxVal = 0; bool any = false;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (mask && !mask[(size_t)y * mstride + x]) continue;
            double v = eig[(size_t)y * w + x];
            if (!any || v > maxVal) { maxVal = v; any = true; }
        }
    if (!any) maxVal = 0;
    const float thr = (float)(maxVal * quality);
    std::vector<GfttCand> cand;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            float val = eig[(size_t)y * w + x];
            if (!(val > thr)) continue;                     // THRESH_TOZERO then val != 0
            if (val == 0.f) continue;
            if (mask && !mask[(size_t)y * mstride + x]) continue;
            bool ismax = true;                              // val == dilate3x3(thresholded eig)
            for (int j = -1; j <= 1 && ismax; j++)
                for (int i = -1; i <= 1; i++) {
                    float nv = eig[(size_t)(y + j) * w + (x + i)];
                    if (nv > thr && nv > val) { ismax = false; break; }
                }
            if (ismax) cand.push_back({val, y * w + x});
        }
    if (ncand_out) *ncand_out = (int)cand.size();
    if (cand.empty()) return 0;
    std::sort(cand.begin(), cand.end(), [](const GfttCand& a, const GfttCand& b) {        // greaterThanPtr
        return (a.val > b.val) ? true : (a.val < b.val) ? false : (a.addr > b.addr);
    });
    int ncorners = 0;
    if (minDistance >= 1) {
        const int cell_size = cvRoundD(minDistance);
        const int grid_width = (w + cell_size - 1) / cell_size;
        const int grid_height = (h + cell_size - 1) / cell_size;
        std::vector<std::vector<std::pair<float, float>>> grid((size_t)grid_width * grid_height);
        minDistance *= minDistance;
        for (size_t i = 0; i < cand.size(); i++) {
            int y = cand[i].addr / w, x = cand[i].addr - y * w;
            bool good = true;
            int x_cell = x / cell_size, y_cell = y / cell_size;
            int x1 = std::max(0, x_cell - 1), y1 = std::max(0, y_cell - 1);
            int x2 = std::min(grid_width - 1, x_cell + 1), y2 = std::min(grid_height - 1, y_cell + 1);
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++) {
                    auto& m = grid[(size_t)yy * grid_width + xx];
                    for (size_t j = 0; j < m.size(); j++) {
                        float dx = x - m[j].first, dy = y - m[j].second;
                        if (dx * dx + dy * dy < minDistance) { good = false; break; }
                    }
                }
            if (good) {
                grid[(size_t)y_cell * grid_width + x_cell].push_back({(float)x, (float)y});
                if (ncorners < xy_cap) { xy[2 * ncorners] = (float)x; xy[2 * ncorners + 1] = (float)y; }
                ++ncorners;
                if (maxCorners > 0 && ncorners == maxCorners) break;
            }
        }
    } else {
        for (size_t i = 0; i < cand.size(); i++) {
            int y = cand[i].addr / w, x = cand[i].addr - y * w;
            if (ncorners < xy_cap) { xy[2 * ncorners] = (float)x; xy[2 * ncorners + 1] = (float)y; }
            ++ncorners;
            if (maxCorners > 0 && ncorners == maxCorners) break;
        }
    }
    return ncorners;
}

}  // namespace

struct lvi_tracker {
    lvi_tracker_params P;
    // FeatureTracker::{cur_img, forw_img} as pyramids (feature_tracker.cpp:94-101, 200-204)
    std::vector<Image8> curPyr, forwPyr;
    int curTop = -1, forwTop = -1;
    bool have_forw = false, have_cur = false;
    std::vector<float> cur_xy, forw_xy, err;
    std::vector<uint8_t> status;
    bool have_lk = false;
    std::vector<uint8_t> mask; bool have_mask = false;
    std::vector<float> eig, gftt_xy; int gftt_n = 0, gftt_ncand = 0; bool have_gftt = false, gftt_pending = false;
    bool equalize = false; double clahe_clip = 3.0; int clahe_tx = 8, clahe_ty = 8;       // EQUALIZE (feature_tracker.cpp:86-90)
};

namespace {

void runLK(lvi_tracker* t, const std::vector<Image8>& prevPyr, int prevTop, const std::vector<Image8>& nextPyr, int nextTop,
           const float* prev_xy, int n, float* next_xy, uint8_t* status, float* err)
{
    const lvi_tracker_params& P = t->P;
    int maxLevel = std::min(prevTop, nextTop);
    int maxCount = std::min(std::max(P.lk_max_iters, 0), 100);
    double eps = std::min(std::max((double)P.lk_eps, 0.), 10.);
    eps *= eps;
    for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0.f; next_xy[2 * i] = 0.f; next_xy[2 * i + 1] = 0.f; }
    Deriv16 dI;
    for (int level = maxLevel; level >= 0; level--) {
        scharr(prevPyr[level], dI);
        lkLevel(prevPyr[level], dI, nextPyr[level], prev_xy, next_xy, status, err, n, P.lk_win, level, maxLevel, maxCount, eps, P.lk_min_eig_threshold);
    }
}

// ---------------------------------------------------------------------------
// f-2  cv::CLAHE::apply for CV_8UC1 (OpenCV 4.x imgproc/src/clahe.cpp: CLAHE_CalcLut_Body, CLAHE_Interpolation_Body;
// source not in the reference tree, restated from the published algorithm — PARITY UNPINNED).
//   * images whose size is not a multiple of the tile grid are extended to the right / bottom with REFLECT_101 for the
//     histograms only;  * clipLimit = max(int(clip * tileArea / 256), 1);  * excess is redistributed evenly, the remainder
//     one count every max(256 / residual, 1) bins;  * lut = saturate(cvRound(cumsum * (255.f / tileArea)));
//   * output = bilinear blend of the four neighbouring tile LUTs, in f32, ((a*xa1 + b*xa)*ya1 + (c*xa1 + d*xa)*ya).
// ---------------------------------------------------------------------------
void clahe(const Image8& src, Image8& dst, double clipLimitD, int tilesX, int tilesY)
{
    const int histSize = 256;
    const int W = src.w, H = src.h;
    int extW = W, extH = H;
    if (W % tilesX != 0 || H % tilesY != 0) { extW = W + (tilesX - (W % tilesX)); extH = H + (tilesY - (H % tilesY)); }   // copyMakeBorder(… tiles - (size % tiles) …)
    const int tw = extW / tilesX, th = extH / tilesY;
    const int tileSizeTotal = tw * th;
    const float lutScale = static_cast<float>(histSize - 1) / tileSizeTotal;
    int clipLimit = 0;
    if (clipLimitD > 0.0) { clipLimit = static_cast<int>(clipLimitD * tileSizeTotal / histSize); clipLimit = std::max(clipLimit, 1); }
    std::vector<uint8_t> lut((size_t)tilesX * tilesY * histSize);
    for (int k = 0; k < tilesX * tilesY; k++) {
        const int ty = k / tilesX, tx = k % tilesX;
        int tileHist[256] = {0};
        for (int y = ty * th; y < (ty + 1) * th; y++)
            for (int x = tx * tw; x < (tx + 1) * tw; x++) {
                // extended image = copyMakeBorder(src, 0, extH - H, 0, extW - W, BORDER_REFLECT_101)
                const int sx = x < W ? x : reflect101(x, W), sy = y < H ? y : reflect101(y, H);
                tileHist[src.at(sx, sy)]++;
            }
        if (clipLimit > 0) {
            int clipped = 0;
            for (int i = 0; i < histSize; ++i)
                if (tileHist[i] > clipLimit) { clipped += tileHist[i] - clipLimit; tileHist[i] = clipLimit; }
            int redistBatch = clipped / histSize;
            int residual = clipped - redistBatch * histSize;
            for (int i = 0; i < histSize; ++i) tileHist[i] += redistBatch;
            if (residual != 0) {
                int residualStep = std::max(histSize / residual, 1);
                for (int i = 0; i < histSize && residual > 0; i += residualStep, residual--) tileHist[i]++;
            }
        }
        int sum = 0;
        uint8_t* tileLut = &lut[(size_t)k * histSize];
        for (int i = 0; i < histSize; ++i) {
            sum += tileHist[i];
            const int v = cvRound(sum * lutScale);
            tileLut[i] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
    dst.w = W; dst.h = H; dst.px.resize((size_t)W * H);
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < H; y++) {
        const float tyf = y * inv_th - 0.5f;
        int ty1 = cvFloor(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        ty1 = std::max(ty1, 0); ty2 = std::min(ty2, tilesY - 1);
        const uint8_t* lutPlane1 = &lut[(size_t)ty1 * tilesX * histSize];
        const uint8_t* lutPlane2 = &lut[(size_t)ty2 * tilesX * histSize];
        for (int x = 0; x < W; x++) {
            const float txf = x * inv_tw - 0.5f;
            int tx1 = cvFloor(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            tx1 = std::max(tx1, 0); tx2 = std::min(tx2, tilesX - 1);
            const int srcVal = src.at(x, y);
            const int ind1 = tx1 * histSize + srcVal, ind2 = tx2 * histSize + srcVal;
            const float res = (lutPlane1[ind1] * xa1 + lutPlane1[ind2] * xa) * ya1 + (lutPlane2[ind1] * xa1 + lutPlane2[ind2] * xa) * ya;
            const int v = cvRound(res);
            dst.px[(size_t)y * W + x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

// ---------------------------------------------------------------------------
// f-3  CataCamera::liftProjective (CataCamera.cc:556-626) with the recursive distortion model (n = 8,
// distortion() :766-783), all in double, followed by undistortedPoints' (b.x/b.z, b.y/b.z) → Point2f
// (feature_tracker.cpp:306-309).
// ---------------------------------------------------------------------------
inline void mei_distortion(const lvi_mei_params& c, double pux, double puy, double& dux, double& duy)
{
    const double k1 = c.k1, k2 = c.k2, p1 = c.p1, p2 = c.p2;
    const double mx2_u = pux * pux, my2_u = puy * puy, mxy_u = pux * puy;
    const double rho2_u = mx2_u + my2_u;
    const double rad_dist_u = k1 * rho2_u + k2 * rho2_u * rho2_u;
    dux = pux * rad_dist_u + 2.0 * p1 * mxy_u + p2 * (rho2_u + 2.0 * mx2_u);
    duy = puy * rad_dist_u + 2.0 * p2 * mxy_u + p1 * (rho2_u + 2.0 * my2_u);
}
inline void mei_undistort(const lvi_mei_params& c, float px, float py, float& ux, float& uy)
{
    const double inv_K11 = 1.0 / c.gamma1, inv_K13 = -c.u0 / c.gamma1, inv_K22 = 1.0 / c.gamma2, inv_K23 = -c.v0 / c.gamma2;   // :320-323
    const bool noDistortion = c.k1 == 0.0 && c.k2 == 0.0 && c.p1 == 0.0 && c.p2 == 0.0;                                              // :307-317
    const double mx_d = inv_K11 * (double)px + inv_K13, my_d = inv_K22 * (double)py + inv_K23;
    double mx_u, my_u;
    if (noDistortion) { mx_u = mx_d; my_u = my_d; }
    else {
        double dux, duy;
        mei_distortion(c, mx_d, my_d, dux, duy);
        mx_u = mx_d - dux; my_u = my_d - duy;
        for (int i = 1; i < 8; ++i) { mei_distortion(c, mx_u, my_u, dux, duy); mx_u = mx_d - dux; my_u = my_d - duy; }
    }
    double bz;
    const double xi = c.xi;
    if (xi == 1.0) bz = (1.0 - mx_u * mx_u - my_u * my_u) / 2.0;
    else { const double rho2_d = mx_u * mx_u + my_u * my_u; bz = 1.0 - xi * (rho2_d + 1.0) / (xi + std::sqrt(1.0 + (1.0 - xi * xi) * rho2_d)); }
    ux = (float)(mx_u / bz); uy = (float)(my_u / bz);
}

bool load_image(Image8& im, const uint8_t* img, int w, int h, int stride)
{
    im.w = w; im.h = h; im.px.resize((size_t)w * h);
    for (int y = 0; y < h; y++) std::memcpy(&im.px[(size_t)y * w], img + (size_t)y * stride, w);
    return true;
}

}  // namespace

extern "C" {

void lvi_tracker_params_default(lvi_tracker_params* p)
{
    std::memset(p, 0, sizeof(*p));
    p->max_width = 1280; p->max_height = 720;
    p->max_cnt = 150; p->min_dist = 20.0;
    p->lk_win = 21; p->lk_max_level = 3; p->lk_max_iters = 30; p->lk_eps = 0.01; p->lk_min_eig_threshold = 1e-4f;
    p->gftt_quality = 0.01;
    p->max_features = 1024;
}

int32_t lvi_tracker_create(const lvi_tracker_params* p, int32_t, lvi_tracker** out)
{
    if (!p || !out) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    if (p->lk_win < 3 || (p->lk_win & 1) == 0 || p->lk_max_level < 0 || p->lk_max_level > 7) return tfail(LVI_ERR_INVALID_ARG, "bad LK parameters");
    lvi_tracker* t = new lvi_tracker();
    t->P = *p;
    *out = t;
    return LVI_OK;
}
void lvi_tracker_destroy(lvi_tracker* t) { delete t; }
int32_t lvi_tracker_sync(lvi_tracker*) { return LVI_OK; }

int32_t lvi_tracker_push_image(lvi_tracker* t, const uint8_t* img, int32_t w, int32_t h, int32_t stride)
{
    if (!t || !img || w <= 0 || h <= 0 || stride < w) return tfail(LVI_ERR_INVALID_ARG, "bad image");
    if (w > t->P.max_width || h > t->P.max_height) return tfail(LVI_ERR_CAPACITY, "image exceeds capacity");
    if (t->have_forw) { t->curPyr.swap(t->forwPyr); t->curTop = t->forwTop; t->have_cur = true; }
    Image8 im; load_image(im, img, w, h, stride);
    if (t->equalize) { Image8 eq; clahe(im, eq, t->clahe_clip, t->clahe_tx, t->clahe_ty); im = eq; }     // :86-90
    t->forwTop = buildPyramid(im, t->forwPyr, t->P.lk_win, t->P.lk_max_level);
    if (!t->have_forw) { t->curPyr = t->forwPyr; t->curTop = t->forwTop; t->have_cur = true; }   // prev = cur = forw = img (:94-97)
    t->have_forw = true; t->have_lk = false; t->have_gftt = false;
    return LVI_OK;
}
int32_t lvi_clahe(lvi_tracker* t, const uint8_t* img, int32_t w, int32_t h, int32_t stride, double clip_limit, int32_t tiles_x, int32_t tiles_y,
                  uint8_t* out, int32_t out_stride)
{
    if (!t || !img || !out || w <= 0 || h <= 0 || stride < w || out_stride < w) return tfail(LVI_ERR_INVALID_ARG, "bad image");
    if (tiles_x < 1 || tiles_y < 1 || tiles_x > 64 || tiles_y > 64) return tfail(LVI_ERR_INVALID_ARG, "bad tile grid");
    if (w > t->P.max_width || h > t->P.max_height) return tfail(LVI_ERR_CAPACITY, "image exceeds capacity");
    Image8 im, eq; load_image(im, img, w, h, stride);
    clahe(im, eq, clip_limit, tiles_x, tiles_y);
    for (int y = 0; y < h; y++) std::memcpy(out + (size_t)y * out_stride, &eq.px[(size_t)y * w], w);
    return LVI_OK;
}
int32_t lvi_tracker_set_equalize(lvi_tracker* t, int32_t on, double clip_limit, int32_t tiles_x, int32_t tiles_y)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null handle");
    if (on && (tiles_x < 1 || tiles_y < 1 || tiles_x > 64 || tiles_y > 64)) return tfail(LVI_ERR_INVALID_ARG, "bad tile grid");
    t->equalize = on != 0; t->clahe_clip = clip_limit; t->clahe_tx = tiles_x; t->clahe_ty = tiles_y;
    return LVI_OK;
}
int32_t lvi_undistort_points(lvi_tracker* t, const lvi_mei_params* cam, const float* xy, int32_t n, float* un_xy)
{
    if (!t || !cam || n < 0 || (n > 0 && (!xy || !un_xy))) return tfail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (n > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "too many points");
    for (int i = 0; i < n; i++) mei_undistort(*cam, xy[2 * i], xy[2 * i + 1], un_xy[2 * i], un_xy[2 * i + 1]);
    return LVI_OK;
}
int32_t lvi_tracker_set_points(lvi_tracker* t, const float* cur_xy, int32_t n)
{
    if (!t || n < 0 || (n > 0 && !cur_xy)) return tfail(LVI_ERR_INVALID_ARG, "bad points");
    if (n > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "too many points");
    t->cur_xy.assign(cur_xy, cur_xy + 2 * (size_t)n);
    t->have_lk = false;
    return LVI_OK;
}
int32_t lvi_tracker_run_lk(lvi_tracker* t)
{
    if (!t || !t->have_forw || !t->have_cur) return tfail(LVI_ERR_STATE, "no image pair");
    if (t->curPyr[0].w != t->forwPyr[0].w || t->curPyr[0].h != t->forwPyr[0].h) return tfail(LVI_ERR_INVALID_ARG, "image size changed");
    int n = (int)t->cur_xy.size() / 2;
    t->forw_xy.assign(2 * (size_t)n, 0.f); t->status.assign(n, 1); t->err.assign(n, 0.f);
    runLK(t, t->curPyr, t->curTop, t->forwPyr, t->forwTop, t->cur_xy.data(), n, t->forw_xy.data(), t->status.data(), t->err.data());
    t->have_lk = true;
    return LVI_OK;
}
int32_t lvi_tracker_get_lk(lvi_tracker* t, float* forw_xy, uint8_t* status, float* err, int32_t capacity, int32_t* n)
{
    if (!t || !n) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    if (!t->have_lk) return tfail(LVI_ERR_STATE, "LK not run");
    *n = (int32_t)t->status.size();
    if (capacity < *n) return tfail(LVI_ERR_CAPACITY, "capacity too small");
    if (*n) {
        if (forw_xy) std::memcpy(forw_xy, t->forw_xy.data(), sizeof(float) * 2 * *n);
        if (status) std::memcpy(status, t->status.data(), *n);
        if (err) std::memcpy(err, t->err.data(), sizeof(float) * *n);
    }
    return LVI_OK;
}
int32_t lvi_tracker_set_mask(lvi_tracker* t, const uint8_t* mask, int32_t w, int32_t h, int32_t stride)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    if (!mask) { t->have_mask = false; t->mask.clear(); return LVI_OK; }
    if (!t->have_forw || w != t->forwPyr[0].w || h != t->forwPyr[0].h || stride < w) return tfail(LVI_ERR_INVALID_ARG, "mask size mismatch");
    t->mask.resize((size_t)w * h);
    for (int y = 0; y < h; y++) std::memcpy(&t->mask[(size_t)y * w], mask + (size_t)y * stride, w);
    t->have_mask = true;
    return LVI_OK;
}
int32_t lvi_tracker_run_gftt(lvi_tracker* t, int32_t max_corners)
{
    if (!t || !t->have_forw) return tfail(LVI_ERR_STATE, "no image");
    const Image8& im = t->forwPyr[0];
    t->gftt_xy.assign(2 * (size_t)std::max(1, t->P.max_features), 0.f);
    int n = goodFeatures(im, t->have_mask ? t->mask.data() : nullptr, im.w, max_corners, t->P.gftt_quality, t->P.min_dist,
                         t->eig, t->gftt_xy.data(), t->P.max_features, &t->gftt_ncand);
    if (n > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "more corners than max_features");
    t->gftt_n = n; t->have_gftt = true;
    return LVI_OK;
}
int32_t lvi_tracker_get_gftt(lvi_tracker* t, float* xy, int32_t capacity, int32_t* n)
{
    if (!t || !n) return tfail(LVI_ERR_INVALID_ARG, "null argument");
    if (!t->have_gftt) return tfail(LVI_ERR_STATE, "GFTT not run");
    *n = t->gftt_n;
    if (capacity < *n) return tfail(LVI_ERR_CAPACITY, "capacity too small");
    if (*n && xy) std::memcpy(xy, t->gftt_xy.data(), sizeof(float) * 2 * *n);
    return LVI_OK;
}

// cv::circle(mask, pt, radius, 0, -1) as OpenCV's FillCircle rasterises it (imgproc/src/drawing.cpp): the midpoint recurrence,
// horizontal spans clipped to the image (feature_tracker.cpp:64-66)
static void fillCircleZeroMask(std::vector<uint8_t>& img, int w, int h, int cx, int cy, int radius)
{
    auto hline = [&](int y, int x0, int x1) {
        if (y < 0 || y >= h) return;
        x0 = std::max(x0, 0); x1 = std::min(x1, w - 1);
        for (int x = x0; x <= x1; x++) img[(size_t)y * w + x] = 0;
    };
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hline(cy - dy, cx - dx, cx + dx); hline(cy + dy, cx - dx, cx + dx);
        hline(cy - dx, cx - dy, cx + dy); hline(cy + dx, cx - dy, cx + dy);
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}
int32_t lvi_tracker_set_mask_circles(lvi_tracker* t, const float* centers_xy, int32_t n, int32_t radius)
{
    if (!t || n < 0 || (n > 0 && !centers_xy) || radius < 0 || radius > 120) return tfail(LVI_ERR_INVALID_ARG, "bad circle list");
    if (!t->have_forw) return tfail(LVI_ERR_STATE, "no image");
    if (n > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "too many circles");
    const int w = t->forwPyr[0].w, h = t->forwPyr[0].h;
    t->mask.assign((size_t)w * h, 255);
    for (int i = 0; i < n; i++) fillCircleZeroMask(t->mask, w, h, cvRound(centers_xy[2 * i]), cvRound(centers_xy[2 * i + 1]), radius);
    t->have_mask = true;
    return LVI_OK;
}
int32_t lvi_tracker_run_gftt_async(lvi_tracker* t, int32_t max_corners)
{
    const int32_t st = lvi_tracker_run_gftt(t, max_corners);
    if (st == LVI_OK) t->gftt_pending = true;
    return st;
}
int32_t lvi_tracker_finish_frame(lvi_tracker* t, const lvi_mei_params* cam, const float* kept_xy, int32_t n_kept,
                                 float* new_xy, int32_t new_capacity, int32_t* n_new, float* un_xy)
{
    if (!t || n_kept < 0 || (n_kept > 0 && !kept_xy) || !n_new) return tfail(LVI_ERR_INVALID_ARG, "bad arguments");
    if (n_kept > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "too many points");
    const int nn = t->gftt_pending ? t->gftt_n : 0;
    t->gftt_pending = false;
    if (n_kept + nn > t->P.max_features) return tfail(LVI_ERR_CAPACITY, "more corners than max_features");
    if (nn > new_capacity) return tfail(LVI_ERR_CAPACITY, "capacity too small");
    *n_new = nn;
    if (nn && new_xy) std::memcpy(new_xy, t->gftt_xy.data(), sizeof(float) * 2 * (size_t)nn);
    if (cam && un_xy) {
        for (int i = 0; i < n_kept; i++) mei_undistort(*cam, kept_xy[2 * i], kept_xy[2 * i + 1], un_xy[2 * i], un_xy[2 * i + 1]);
        for (int i = 0; i < nn; i++) mei_undistort(*cam, t->gftt_xy[2 * i], t->gftt_xy[2 * i + 1], un_xy[2 * (n_kept + i)], un_xy[2 * (n_kept + i) + 1]);
    }
    return LVI_OK;
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
    double q0 = t->P.gftt_quality, d0 = t->P.min_dist;
    t->P.gftt_quality = quality; t->P.min_dist = min_dist;
    st = lvi_tracker_run_gftt(t, max_corners);
    t->P.gftt_quality = q0; t->P.min_dist = d0;
    if (st) return st;
    return lvi_tracker_get_gftt(t, xy, xy_capacity, n_out);
}

int32_t lvi_tracker_debug_get(lvi_tracker* t, int32_t what, void* dst, int64_t cap, int64_t* n_bytes)
{
    if (!t) return tfail(LVI_ERR_INVALID_ARG, "null handle");
    const void* src = nullptr; int64_t bytes = 0;
    switch (what) {
        case LVI_TDBG_PYRAMID_L1: case LVI_TDBG_PYRAMID_L2: case LVI_TDBG_PYRAMID_L3: {
            int l = what - LVI_TDBG_PYRAMID_L1 + 1;
            if (!t->have_forw || l > t->forwTop) return tfail(LVI_ERR_STATE, "level not built");
            src = t->forwPyr[l].px.data(); bytes = (int64_t)t->forwPyr[l].px.size();
            break;
        }
        case LVI_TDBG_MINEIG:
            if (!t->have_gftt) return tfail(LVI_ERR_STATE, "GFTT not run");
            src = t->eig.data(); bytes = (int64_t)(t->eig.size() * sizeof(float));
            break;
        case LVI_TDBG_GFTT_NCAND:
            if (!t->have_gftt) return tfail(LVI_ERR_STATE, "GFTT not run");
            src = &t->gftt_ncand; bytes = sizeof(int32_t);
            break;
        case LVI_TDBG_MASK:
            if (!t->have_mask || !t->have_forw) return tfail(LVI_ERR_STATE, "no mask");
            src = t->mask.data(); bytes = (int64_t)t->mask.size();
            break;
        default: return tfail(LVI_ERR_INVALID_ARG, "unknown debug item");
    }
    if (n_bytes) *n_bytes = bytes;
    if (!dst) return LVI_OK;
    if (cap < bytes) return tfail(LVI_ERR_CAPACITY, "debug buffer too small");
    std::memcpy(dst, src, (size_t)bytes);
    return LVI_OK;
}

int32_t lvi_tracker_prof_enable(lvi_tracker*, int32_t) { return LVI_OK; }
int32_t lvi_tracker_prof_reset(lvi_tracker*) { return LVI_OK; }
int32_t lvi_tracker_prof_read(lvi_tracker*, lvi_kernel_stat*, int32_t, int32_t* n) { if (n) *n = 0; return LVI_OK; }

}  // extern "C"
