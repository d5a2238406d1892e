// Microbenchmark (not part of the product): throughput of direct-addressed voxel accumulation with
// global u64 atomics on gfx950, with and without wave-level merging of runs of equal keys.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Cell { unsigned long long sx, sy, sz, si; unsigned int n; unsigned int pad[3]; };   // 48 B

__global__ void acc_plain(const float4* __restrict__ p, int n, Cell* __restrict__ tab, float inv, int dx, int dxy)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 q = p[i];
    int ix = (int)floorf(q.x * inv), iy = (int)floorf(q.y * inv), iz = (int)floorf(q.z * inv);
    int key = ix + iy * dx + iz * dxy;
    Cell* c = tab + key;
    atomicAdd(&c->sx, (unsigned long long)((q.x * inv - ix) * 1048576.f));
    atomicAdd(&c->sy, (unsigned long long)((q.y * inv - iy) * 1048576.f));
    atomicAdd(&c->sz, (unsigned long long)((q.z * inv - iz) * 1048576.f));
    atomicAdd(&c->si, (unsigned long long)(q.w * 65536.f));
    atomicAdd(&c->n, 1u);
}

__global__ void acc_merge(const float4* __restrict__ p, int n, Cell* __restrict__ tab, float inv, int dx, int dxy)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lane = threadIdx.x & 63;
    bool live = i < n;
    float4 q = live ? p[i] : make_float4(0, 0, 0, 0);
    int ix = (int)floorf(q.x * inv), iy = (int)floorf(q.y * inv), iz = (int)floorf(q.z * inv);
    int key = live ? ix + iy * dx + iz * dxy : -1;
    unsigned long long ax = (unsigned long long)((q.x * inv - ix) * 1048576.f), ay = (unsigned long long)((q.y * inv - iy) * 1048576.f),
                       az = (unsigned long long)((q.z * inv - iz) * 1048576.f), ai = (unsigned long long)(q.w * 65536.f);
    unsigned int cnt = 1;
    // segmented inclusive scan over runs of equal consecutive keys
    int prev = __shfl_up(key, 1);
    bool head = lane == 0 || prev != key;
    unsigned long long hm = __ballot(head);
    // distance to own run head
    unsigned long long below = hm & ((2ull << lane) - 1ull);
    int hpos = 63 - __clzll(below);
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long bx = __shfl_up(ax, d), by = __shfl_up(ay, d), bz = __shfl_up(az, d), bi = __shfl_up(ai, d);
        unsigned int bc = __shfl_up(cnt, d);
        if (lane - d >= hpos) { ax += bx; ay += by; az += bz; ai += bi; cnt += bc; }
    }
    int nxt = __shfl_down(key, 1);
    bool tail = lane == 63 || nxt != key;
    if (live && tail) {
        Cell* c = tab + key;
        atomicAdd(&c->sx, ax); atomicAdd(&c->sy, ay); atomicAdd(&c->sz, az); atomicAdd(&c->si, ai); atomicAdd(&c->n, cnt);
    }
}

__global__ void count_tails(const float4* __restrict__ p, int n, float inv, int dx, int dxy, unsigned int* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 q = p[i];
    int key = (int)floorf(q.x * inv) + (int)floorf(q.y * inv) * dx + (int)floorf(q.z * inv) * dxy;
    int nxt = __shfl_down(key, 1);
    if ((threadIdx.x & 63) == 63 || nxt != key) atomicAdd(out, 1u);
}

int main(int argc, char** argv)
{
    int n = 5000000;
    float step = argc > 1 ? atof(argv[1]) : 0.03f;
    float leaf = argc > 2 ? atof(argv[2]) : 0.4f;
    float X = 40, Y = 30, Z = 10;
    int dx = (int)(X / leaf) + 1, dy = (int)(Y / leaf) + 1, dz = (int)(Z / leaf) + 1;
    size_t cells = (size_t)dx * dy * dz;
    std::vector<float4> h(n);
    srand(1);
    float x = 0, y = 0, z = 0, ux = 1, uy = 0, uz = 0;
    for (int i = 0; i < n; i++) {
        if (i % 2000 == 0) { x = X * (rand() / (float)RAND_MAX); y = Y * (rand() / (float)RAND_MAX); z = Z * (rand() / (float)RAND_MAX);
            float a = 6.28f * rand() / RAND_MAX; ux = cosf(a); uy = sinf(a); uz = 0.1f * (rand() / (float)RAND_MAX - 0.5f); }
        x += ux * step; y += uy * step; z += uz * step;
        if (x < 0 || x >= X) { ux = -ux; x += 2 * ux * step; } if (y < 0 || y >= Y) { uy = -uy; y += 2 * uy * step; } if (z < 0 || z >= Z) { uz = -uz; z += 2 * uz * step; }
        h[i] = make_float4(fminf(fmaxf(x, 0.f), X - 1e-3f), fminf(fmaxf(y, 0.f), Y - 1e-3f), fminf(fmaxf(z, 0.f), Z - 1e-3f), (float)(i & 255));
    }
    float4* d; Cell* tab; unsigned int* dcnt;
    CK(hipMalloc(&d, n * sizeof(float4))); CK(hipMalloc(&tab, cells * sizeof(Cell))); CK(hipMalloc(&dcnt, 4));
    CK(hipMemcpy(d, h.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    CK(hipMemset(dcnt, 0, 4));
    int nb = (n + 255) / 256;
    count_tails<<<nb, 256>>>(d, n, 1.f / leaf, dx, dx * dy, dcnt);
    unsigned int tails; CK(hipMemcpy(&tails, dcnt, 4, hipMemcpyDeviceToHost));
    printf("n=%d cells=%zu (%.1f MB table) step=%.3f leaf=%.2f  wave-run tails=%u (avg run %.2f)\n", n, cells, cells * sizeof(Cell) / 1e6, step, leaf, tails, n / (double)tails);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; variant++) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipMemsetAsync(tab, 0, cells * sizeof(Cell)));
            CK(hipEventRecord(e0));
            if (variant == 0) acc_plain<<<nb, 256>>>(d, n, tab, 1.f / leaf, dx, dx * dy);
            else if (variant == 1) acc_merge<<<nb, 256>>>(d, n, tab, 1.f / leaf, dx, dx * dy);
            else CK(hipMemsetAsync(tab, 0, cells * sizeof(Cell)));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms;
        }
        printf("  %s: %.1f us\n", variant == 0 ? "plain 5 atomics/pt" : variant == 1 ? "wave run-merge     " : "table clear        ", best * 1e3f);
    }
    return 0;
}
