// Native multi-GPU replay (one process, one handle per GPU, RCCL over xGMI): independent scans against a frozen raw local map,
// scan i -> GPU i mod N, every GPU holds its own replica of the map, and after every step the 32-byte pose records of the step are
// all-gathered across the GPUs with ONE grouped ncclAllGather (ncclCommInitAll communicators: no MPI, no launcher).  Nothing else is
// exchanged: the algorithm has no collective inside it.  The same sharding as bench.py / replay.py, without Python.
//
//   replay_multi <n_gpus> <Horizon_SCAN> <scans.bin> <n_scans> <n_raw> <map_corner.bin> <nc> <map_surf.bin> <ns> <guesses.bin> <icp_iters>
//
// scans.bin: n_scans x n_raw lvi_livox_pt; guesses.bin: n_scans x 6 float.  Prints one line per scan: "rec <i> <status> <iters> <pose x6>",
// read back from GPU 0's gathered table.  Built by hand (hipcc … -llvi_hip -lrccl); tests/test_gpu_rccl.py runs it with the GPUs the box
// has (one): what has executed of it is the one-rank form — see DESIGN 7.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "lvi_hotpath.h"

#define CK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)
#define CK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, ncclGetErrorString(r_), __FILE__, __LINE__); exit(4); } } while (0)
#define CK_LVI(x) do { int32_t s_ = (x); if (s_ < 0) { fprintf(stderr, "%s: %d %s (%s:%d)\n", #x, s_, lvi_last_error(), __FILE__, __LINE__); exit(5); } } while (0)

template <class T>
static std::vector<T> read_file(const char* path, size_t n)
{
    std::vector<T> v(n);
    std::ifstream f(path, std::ios::binary);
    if (!f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)))) { fprintf(stderr, "short read: %s\n", path); exit(2); }
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 12) { fprintf(stderr, "usage: replay_multi n_gpus Horizon_SCAN scans.bin n_scans n_raw mc.bin nc ms.bin ns guesses.bin icp_iters\n"); return 2; }
    int ndev = 0;
    CK_HIP(hipGetDeviceCount(&ndev));
    const int N = std::min(std::max(atoi(argv[1]), 1), ndev);
    const int n_scans = atoi(argv[4]), n_raw = atoi(argv[5]), nc = atoi(argv[7]), ns = atoi(argv[9]);
    auto scans = read_file<lvi_livox_pt>(argv[3], (size_t)n_scans * n_raw);
    auto mc = read_file<lvi_pt>(argv[6], nc);
    auto ms = read_file<lvi_pt>(argv[8], ns);
    auto guesses = read_file<float>(argv[10], (size_t)n_scans * 6);
    const int steps = (n_scans + N - 1) / N;

    std::vector<int> devs(N);
    for (int d = 0; d < N; d++) devs[d] = d;
    std::vector<ncclComm_t> comm(N);
    CK_NCCL(ncclCommInitAll(comm.data(), N, devs.data()));          // RCCL: one communicator per GPU, single process

    std::vector<lvi_lidar*> h(N, nullptr);
    std::vector<hipStream_t> cs(N);
    std::vector<lvi_livox_pt*> d_scan(N, nullptr);
    std::vector<float*> d_mine(N, nullptr), d_all(N, nullptr);       // this GPU's record of the step | the step's records of all GPUs, per step
    for (int d = 0; d < N; d++) {
        CK_HIP(hipSetDevice(d));
        lvi_lidar_params P; lvi_lidar_params_default(&P);
        P.Horizon_SCAN = atoi(argv[2]); P.max_raw_points = n_raw + 16; P.max_map_points = std::max(nc, ns) + 16;
        P.icp_max_iters = atoi(argv[11]);
        CK_LVI(lvi_lidar_create(&P, d, &h[d]));
        CK_LVI(lvi_map_upload(h[d], mc.data(), nc, ms.data(), ns));  // every GPU its own replica of the frozen raw map
        CK_LVI(lvi_map_build(h[d]));
        CK_HIP(hipStreamCreateWithFlags(&cs[d], hipStreamNonBlocking));
        CK_HIP(hipMalloc((void**)&d_scan[d], sizeof(lvi_livox_pt) * (size_t)n_raw));
        CK_HIP(hipMalloc((void**)&d_mine[d], sizeof(float) * 8 * (size_t)steps));
        CK_HIP(hipMalloc((void**)&d_all[d], sizeof(float) * 8 * (size_t)steps * N));
        CK_HIP(hipMemset(d_mine[d], 0, sizeof(float) * 8 * (size_t)steps));
    }
    for (int s = 0; s < steps; s++) {
        for (int d = 0; d < N; d++) {
            const int i = s * N + d;
            CK_HIP(hipSetDevice(d));
            if (i >= n_scans) {                                        // a padding record: status -999
                float pad[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int st = -999; memcpy(&pad[6], &st, 4);
                CK_HIP(hipMemcpy(d_mine[d] + 8 * (size_t)s, pad, sizeof(pad), hipMemcpyHostToDevice));
                continue;
            }
            CK_HIP(hipMemcpy(d_scan[d], scans.data() + (size_t)i * n_raw, sizeof(lvi_livox_pt) * (size_t)n_raw, hipMemcpyHostToDevice));
            CK_LVI(lvi_scan_upload_device(h[d], d_scan[d], n_raw));
            CK_LVI(lvi_scan_organize(h[d])); CK_LVI(lvi_scan_extract(h[d])); CK_LVI(lvi_scan_downsample(h[d]));
            CK_LVI(lvi_scan_match_async(h[d], guesses.data() + (size_t)i * 6, d_mine[d] + 8 * (size_t)s));
        }
        for (int d = 0; d < N; d++) { CK_HIP(hipSetDevice(d)); CK_LVI(lvi_lidar_sync(h[d])); }      // the records of the step are final
        CK_NCCL(ncclGroupStart());
        for (int d = 0; d < N; d++)
            CK_NCCL(ncclAllGather(d_mine[d] + 8 * (size_t)s, d_all[d] + 8 * (size_t)s * N, 8, ncclFloat, comm[d], cs[d]));
        CK_NCCL(ncclGroupEnd());
    }
    for (int d = 0; d < N; d++) { CK_HIP(hipSetDevice(d)); CK_HIP(hipStreamSynchronize(cs[d])); }
    std::vector<float> all((size_t)steps * N * 8);
    CK_HIP(hipSetDevice(0));
    CK_HIP(hipMemcpy(all.data(), d_all[0], sizeof(float) * all.size(), hipMemcpyDeviceToHost));
    printf("backend %s gpus %d steps %d\n", lvi_backend(), N, steps);
    for (int i = 0; i < n_scans; i++) {
        const float* r = &all[(size_t)i * 8];
        int st, it; memcpy(&st, &r[6], 4); memcpy(&it, &r[7], 4);
        printf("rec %d %d %d %.9g %.9g %.9g %.9g %.9g %.9g\n", i, st, it, r[0], r[1], r[2], r[3], r[4], r[5]);
    }
    for (int d = 0; d < N; d++) {
        CK_HIP(hipSetDevice(d));
        lvi_lidar_destroy(h[d]);
        CK_HIP(hipFree(d_scan[d])); CK_HIP(hipFree(d_mine[d])); CK_HIP(hipFree(d_all[d]));
        CK_HIP(hipStreamDestroy(cs[d]));
        ncclCommDestroy(comm[d]);
    }
    return 0;
}
