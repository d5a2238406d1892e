// TEST INFRASTRUCTURE — direct entry points to the oracle's small-matrix restatements so that
// tests/test_oracle_math.py can check them against independent ground truth (numpy/LAPACK).
#include <cstring>

#include "lvo_kdtree.h"
#include "lvo_math.h"

extern "C" {

void lvo_test_jacobi(int n, const float* A, float* W, float* V)
{
    float a[36];
    std::memcpy(a, A, sizeof(float) * n * n);
    lvo::jacobi_f32(a, n, W, V);
}
int lvo_test_solve_qr6(const float* A, const float* b, float* x)
{
    float a[36];
    std::memcpy(a, A, sizeof(a));
    std::memcpy(x, b, sizeof(float) * 6);
    return lvo::solve_qr_f32(a, 6, x) ? 1 : 0;
}
int lvo_test_inv6(const float* A, float* inv) { return lvo::inv_lu_f32(A, 6, inv) ? 1 : 0; }
void lvo_test_colpiv_5x3(const float* A, const float* b, float* x)
{
    float a[5][3];
    std::memcpy(a, A, sizeof(a));
    lvo::colpiv_qr_solve_5x3(a, b, x);
}
void lvo_test_get_transformation(float x, float y, float z, float roll, float pitch, float yaw, float* m12)
{
    lvo::Affine3f t = lvo::getTransformation(x, y, z, roll, pitch, yaw);
    std::memcpy(m12, t.m, sizeof(float) * 12);
}
// transformUpdate's slerp of a pure-roll (axis 0) or pure-pitch (axis 1) rotation towards the IMU value
double lvo_test_slerp_axis(int axis, double from, double to, double w)
{
    lvo::Quat a = axis == 0 ? lvo::quat_setRPY(from, 0, 0) : lvo::quat_setRPY(0, from, 0);
    lvo::Quat b = axis == 0 ? lvo::quat_setRPY(to, 0, 0) : lvo::quat_setRPY(0, to, 0);
    double r, p, y;
    lvo::quat_getRPY(lvo::quat_slerp(a, b, w), r, p, y);
    return axis == 0 ? r : p;
}
// exact 5-NN through the kd-tree restatement: returns found count
int lvo_test_kdtree_knn(const float* xyz, int n, const float* q, int nq, int* idx, float* sqd)
{
    lvo::KdTree3f kd;
    kd.build(xyz, 3, n);
    int total = 0;
    for (int i = 0; i < nq; i++) total += kd.knn(q + 3 * i, 5, idx + 5 * i, sqd + 5 * i);
    return total;
}

}  // extern "C"
