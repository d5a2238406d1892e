// TEST INFRASTRUCTURE — CPU oracle of the hot path.  Not shipped, not linked by the
// product library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it.
//
// PARITY UNPINNED: the arithmetic restated in this file lives in third-party
// libraries that are NOT vendored in the reference tree and are not installed in
// the build container (OpenCV 4.x cv::eigen / cv::solve / gemm / Mat::inv,
// Eigen 3.4 ColPivHouseholderQR, tf2 Quaternion/Matrix3x3, PCL 1.12
// getTransformation).  The reference pins no versions beyond
// `find_package(OpenCV 4 REQUIRED)` and has no tests or golden vectors.  What is
// below restates the published algorithms; it is checked by known-answer tests
// (tests/test_oracle_math.py), not by reference outputs.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>

namespace lvo {

// ---------------------------------------------------------------------------
// pcl::getTransformation(x,y,z,roll,pitch,yaw) (PCL common/eigen.hpp), f32.
// Call sites: mapOptimization.cpp:354,401,406.  Row-major 3x4.
// ---------------------------------------------------------------------------
struct Affine3f { float m[3][4]; };

inline Affine3f getTransformation(float x, float y, float z, float roll, float pitch, float yaw)
{
    float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch),
          E = std::cos(roll), F = std::sin(roll), DE = D * E, DF = D * F;
    Affine3f t;
    t.m[0][0] = A * C; t.m[0][1] = A * DF - B * E; t.m[0][2] = B * F + A * DE; t.m[0][3] = x;
    t.m[1][0] = B * C; t.m[1][1] = A * E + B * DF; t.m[1][2] = B * DE - A * F; t.m[1][3] = y;
    t.m[2][0] = -D;    t.m[2][1] = C * F;          t.m[2][2] = C * E;          t.m[2][3] = z;
    return t;
}

// trans2Affine3f (mapOptimization.cpp:404-407): transformIn = [roll,pitch,yaw,x,y,z]
inline Affine3f trans2Affine3f(const float T[6]) { return getTransformation(T[3], T[4], T[5], T[0], T[1], T[2]); }

// ---------------------------------------------------------------------------
// cv::eigen for a symmetric f32 matrix: OpenCV's Jacobi (modules/core/src/lapack.cpp
// JacobiImpl_).  Eigenvalues descending in W, eigenvectors as ROWS of V.
// A is destroyed.  n <= 6.
// ---------------------------------------------------------------------------
inline void jacobi_f32(float* A, int n, float* W, float* V)
{
    const float eps = std::numeric_limits<float>::epsilon();
    int indR[8], indC[8];
    int i, j, k, m;
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) V[i * n + j] = 0.f; V[i * n + i] = 1.f; }
    int iters, maxIters = n * n * 30;
    float mv = 0.f;
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = std::abs(A[n * k + m]), i = k + 2; i < n; i++) {
                float val = std::abs(A[n * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = std::abs(A[k]), i = 1; i < k; i++) {
                float val = std::abs(A[n * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1) for (iters = 0; iters < maxIters; iters++) {
        // pivot = largest off-diagonal element
        for (k = 0, mv = std::abs(A[indR[0]]), i = 1; i < n - 1; i++) {
            float val = std::abs(A[n * i + indR[i]]);
            if (mv < val) mv = val, k = i;
        }
        int l = indR[k];
        for (i = 1; i < n; i++) {
            float val = std::abs(A[n * indC[i] + i]);
            if (mv < val) mv = val, k = indC[i], l = i;
        }
        float p = A[n * k + l];
        if (std::abs(p) <= eps) break;
        float y = (float)((W[l] - W[k]) * 0.5);
        float t = std::abs(y) + std::hypot(p, y);
        float s = std::hypot(p, t);
        float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t;
        W[l] += t;
        float a0, b0;
#define LVO_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (i = 0; i < k; i++)     LVO_ROT(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; i++) LVO_ROT(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; i++) LVO_ROT(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; i++)     LVO_ROT(V[n * k + i], V[n * l + i]);
#undef LVO_ROT
        for (j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx + 1, mv = std::abs(A[n * idx + m]), i = idx + 2; i < n; i++) {
                    float val = std::abs(A[n * idx + i]);
                    if (mv < val) mv = val, m = i;
                }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = std::abs(A[idx]), i = 1; i < idx; i++) {
                    float val = std::abs(A[n * i + idx]);
                    if (mv < val) mv = val, m = i;
                }
                indC[idx] = m;
            }
        }
    }
    // sort descending
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) {
            std::swap(W[m], W[k]);
            for (i = 0; i < n; i++) std::swap(V[n * m + i], V[n * k + i]);
        }
    }
}

// ---------------------------------------------------------------------------
// cv::solve(A, b, x, DECOMP_QR) for a square f32 system: OpenCV hal QRImpl
// (Householder, no pivoting).  A is n x n row-major (destroyed), b length n
// (overwritten with x).  Returns false when a diagonal of R is < eps.
// ---------------------------------------------------------------------------
inline bool solve_qr_f32(float* A, int n, float* b)
{
    const int m = n;
    float vl[16], hF[16];
    const float eps = std::numeric_limits<float>::epsilon() * 10.f;  // FLT_EPSILON*10 in cv::solve's QR call
    for (int l = 0; l < n; l++) {
        int vlSize = m - l;
        float vlNorm = 0.f;
        for (int i = 0; i < vlSize; i++) { vl[i] = A[(l + i) * n + l]; vlNorm += vl[i] * vl[i]; }
        float tmpV = vl[0];
        vl[0] = vl[0] + (vl[0] >= 0.f ? 1.f : -1.f) * std::sqrt(vlNorm);
        vlNorm = std::sqrt(vlNorm + vl[0] * vl[0] - tmpV * tmpV);
        for (int i = 0; i < vlSize; i++) vl[i] /= vlNorm;
        for (int j = l; j < n; j++) {
            float v_lA = 0.f;
            for (int i = l; i < m; i++) v_lA += vl[i - l] * A[i * n + j];
            for (int i = l; i < m; i++) A[i * n + j] -= 2 * vl[i - l] * v_lA;
        }
        hF[l] = vl[0] * vl[0];
        for (int i = 1; i < vlSize; i++) A[(l + i) * n + l] = vl[i] / vl[0];
    }
    for (int l = 0; l < n; l++) {
        vl[0] = 1.f;
        for (int j = 1; j < m - l; j++) vl[j] = A[(j + l) * n + l];
        float v_lB = 0.f;
        for (int i = l; i < m; i++) v_lB += vl[i - l] * b[i];
        for (int i = l; i < m; i++) b[i] -= 2 * vl[i - l] * v_lB * hF[l];
    }
    for (int i = n - 1; i >= 0; i--) {
        for (int j = n - 1; j > i; j--) b[i] -= b[j] * A[i * n + j];
        if (std::abs(A[i * n + i]) < eps) return false;
        b[i] /= A[i * n + i];
    }
    return true;
}

// ---------------------------------------------------------------------------
// cv::Mat::inv() (DECOMP_LU) for n x n f32: OpenCV hal LUImpl on [A | I].
// Returns false if singular (OpenCV then returns a zero matrix).
// ---------------------------------------------------------------------------
inline bool inv_lu_f32(const float* Ain, int n, float* inv)
{
    float A[36];
    std::memcpy(A, Ain, sizeof(float) * n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) inv[i * n + j] = (i == j) ? 1.f : 0.f;
    const float eps = std::numeric_limits<float>::epsilon() * 10.f;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (std::abs(A[j * n + i]) > std::abs(A[k * n + i])) k = j;
        if (std::abs(A[k * n + i]) < eps) { std::fill(inv, inv + n * n, 0.f); return false; }
        if (k != i) {
            for (int j = i; j < n; j++) std::swap(A[i * n + j], A[k * n + j]);
            for (int j = 0; j < n; j++) std::swap(inv[i * n + j], inv[k * n + j]);
        }
        float d = -1 / A[i * n + i];
        for (int j = i + 1; j < n; j++) {
            float alpha = A[j * n + i] * d;
            for (k = i + 1; k < n; k++) A[j * n + k] += alpha * A[i * n + k];
            for (k = 0; k < n; k++) inv[j * n + k] += alpha * inv[i * n + k];
        }
    }
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            float s = inv[i * n + j];
            for (int k = i + 1; k < n; k++) s -= A[i * n + k] * inv[k * n + j];
            inv[i * n + j] = s / A[i * n + i];
        }
    return true;
}

// cv::gemm for CV_32F accumulates each output element in double
// (GEMMSingleMul<float,double>) and stores f32.  C[m x n] = A[m x k] * B[k x n].
inline void gemm_f32_dacc(const float* A, const float* B, float* C, int m, int k, int n)
{
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int t = 0; t < k; t++) s += (double)A[i * k + t] * (double)B[t * n + j];
            C[i * n + j] = (float)s;
        }
}

// ---------------------------------------------------------------------------
// Eigen::ColPivHouseholderQR<Matrix<float,5,3>>::solve(b) — least squares of the
// 5x3 system at mapOptimization.cpp:1128.  Householder with column pivoting on the
// largest remaining column norm; column norms are recomputed directly every step
// (Eigen 3.4 down-dates them; the pivot order is the same except at near-ties).
// ---------------------------------------------------------------------------
inline void colpiv_qr_solve_5x3(const float Ain[5][3], const float bin[5], float x[3])
{
    const int rows = 5, cols = 3;
    float A[5][3], b[5];
    for (int i = 0; i < rows; i++) { for (int j = 0; j < cols; j++) A[i][j] = Ain[i][j]; b[i] = bin[i]; }
    int perm[3] = {0, 1, 2};
    int nonzero_pivots = cols;
    float maxpivot = 0.f;
    float max_norm0 = 0.f;
    for (int j = 0; j < cols; j++) {
        float s = 0.f; for (int i = 0; i < rows; i++) s += A[i][j] * A[i][j];
        max_norm0 = std::max(max_norm0, std::sqrt(s));
    }
    const float thr_helper = (max_norm0 * std::numeric_limits<float>::epsilon()) * (max_norm0 * std::numeric_limits<float>::epsilon()) / (float)rows;
    for (int k = 0; k < cols; k++) {
        // biggest remaining column (squared norm of the tail)
        int big = k; float bigsq = -1.f;
        for (int j = k; j < cols; j++) {
            float s = 0.f; for (int i = k; i < rows; i++) s += A[i][j] * A[i][j];
            if (s > bigsq) { bigsq = s; big = j; }
        }
        if (nonzero_pivots == cols && bigsq < thr_helper * (float)(rows - k)) nonzero_pivots = k;
        if (big != k) { for (int i = 0; i < rows; i++) std::swap(A[i][k], A[i][big]); std::swap(perm[k], perm[big]); }
        // makeHouseholderInPlace on A[k..rows-1][k]
        float c0 = A[k][k];
        float tailSq = 0.f; for (int i = k + 1; i < rows; i++) tailSq += A[i][k] * A[i][k];
        float tau, beta;
        if (tailSq <= std::numeric_limits<float>::min()) {
            tau = 0.f; beta = c0;
            for (int i = k + 1; i < rows; i++) A[i][k] = 0.f;
        } else {
            beta = std::sqrt(c0 * c0 + tailSq);
            if (c0 >= 0.f) beta = -beta;
            for (int i = k + 1; i < rows; i++) A[i][k] /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        if (std::abs(beta) > maxpivot) maxpivot = std::abs(beta);
        // apply H = I - tau v v^T (v = [1; essential]) to the trailing columns and to b
        // (Eigen applyHouseholderOnTheLeft: tmp = essential^T * bottom; tmp += row0)
        for (int j = k + 1; j < cols; j++) {
            float s = 0.f;
            for (int i = k + 1; i < rows; i++) s += A[i][k] * A[i][j];
            s += A[k][j];
            s *= tau;
            A[k][j] -= s;
            for (int i = k + 1; i < rows; i++) A[i][j] -= s * A[i][k];
        }
        {
            float s = 0.f;
            for (int i = k + 1; i < rows; i++) s += A[i][k] * b[i];
            s += b[k];
            s *= tau;
            b[k] -= s;
            for (int i = k + 1; i < rows; i++) b[i] -= s * A[i][k];
        }
    }
    // back substitution on the top-left nonzero_pivots block
    float c[3] = {0.f, 0.f, 0.f};
    for (int i = nonzero_pivots - 1; i >= 0; i--) {
        float s = b[i];
        for (int j = i + 1; j < nonzero_pivots; j++) s -= A[i][j] * c[j];
        c[i] = s / A[i][i];
    }
    x[0] = x[1] = x[2] = 0.f;
    for (int i = 0; i < nonzero_pivots; i++) x[perm[i]] = c[i];
}

// ---------------------------------------------------------------------------
// tf2 (doubles): Quaternion::setRPY, slerp, Matrix3x3(q).getRPY — used only at
// mapOptimization.cpp:1352-1366.
// ---------------------------------------------------------------------------
struct Quat { double x, y, z, w; };

inline Quat quat_setRPY(double roll, double pitch, double yaw)
{
    double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
    Quat q;
    q.x = sr * cp * cy - cr * sp * sy;
    q.y = cr * sp * cy + sr * cp * sy;
    q.z = cr * cp * sy - sr * sp * cy;
    q.w = cr * cp * cy + sr * sp * sy;
    return q;
}
inline double quat_dot(const Quat& a, const Quat& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

inline Quat quat_slerp(const Quat& a, const Quat& q, double t)
{
    double s = std::sqrt(quat_dot(a, a) * quat_dot(q, q));
    double d = quat_dot(a, q);
    double ang = (d < 0) ? std::acos(-d / s) * 2.0 : std::acos(d / s) * 2.0;   // angleShortestPath
    double theta = ang / 2.0;
    if (theta != 0.0) {
        double dd = 1.0 / std::sin(theta);
        double s0 = std::sin((1.0 - t) * theta);
        double s1 = std::sin(t * theta);
        if (d < 0) return Quat{(a.x * s0 + -q.x * s1) * dd, (a.y * s0 + -q.y * s1) * dd, (a.z * s0 + -q.z * s1) * dd, (a.w * s0 + -q.w * s1) * dd};
        return Quat{(a.x * s0 + q.x * s1) * dd, (a.y * s0 + q.y * s1) * dd, (a.z * s0 + q.z * s1) * dd, (a.w * s0 + q.w * s1) * dd};
    }
    return a;
}

inline void quat_getRPY(const Quat& q, double& roll, double& pitch, double& yaw)
{
    double d = quat_dot(q, q);
    double s = 2.0 / d;
    double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    (void)wz;
    if (std::fabs(m20) >= 1) {
        yaw = 0;
        double delta = std::atan2(m21, m22);
        if (m20 < 0) { pitch = M_PI / 2.0; roll = delta; }
        else { pitch = -M_PI / 2.0; roll = delta; }
    } else {
        pitch = -std::asin(m20);
        roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
        yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
    }
}

}  // namespace lvo
