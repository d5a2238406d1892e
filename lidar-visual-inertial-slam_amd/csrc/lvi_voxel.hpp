// pcl::VoxelGrid<PointXYZI>::filter on the GPU, batched over independent segments
// (the 4 per-ring calls of featureExtraction.cpp:239-243, the corner+surf pair of
// mapOptimization.cpp:987-999 and the corner+surf pair of :958-965 are one batch each).
#pragma once
#include "lvi_sort.hpp"

namespace lvi {

struct VoxSegStatic {            // filled on the host once per plan
    const lvi_pt* in;            // input array
    const uint8_t* mask;         // optional: only points with mask != 0 take part (indexed like `in`)
    lvi_pt* out;                 // output array
    float leaf;
};
struct VoxSegDyn { int in_off; int n; };   // device: where the segment's input starts and how long it is

struct VoxGrid {                 // device, per segment
    unsigned bb[6];              // order-encoded min xyz, max xyz
    int n_valid;
    float inv;                   // inverse_leaf_size_
    int min_b[3];
    int div_b[3];
    unsigned mul1, mul2;         // divb_mul_[1], divb_mul_[2]
    unsigned sentinel;           // key of masked-out points, larger than every real key
    int nbits;
    int overflow;                // PCL's "leaf size too small" rule hit: output = input
    int nvox;
    int out_off;
    // ---- centroid arithmetic (every realisation, the incremental map included): exact integer sums of
    //      (value - cell * leaf) * 2^fx_k per voxel, cell = the voxel's ABSOLUTE integer coordinate floor(value * inv): the origin
    //      of a voxel's sums is a property of the voxel, not of the bounding box, so the sums do not depend on which other points
    //      are in the cloud, nor on the order the points of a voxel are visited in.  Intensity: value * 2^fx_ki (origin 0).
    double leaf_d;               // (double)leaf
    int fx_k, fx_ki;             // scale exponents: |value - cell*leaf| < 2 leaf < 2^(37 - fx_k); |intensity| < 2^(37 - fx_ki), at least 2^8
    // ---- binned path: bin = idx >> bin_shift, nbins <= VB_NB
    int bin_shift, nbins;
    unsigned long long ncells;   // div_b product (0 when the overflow rule fired or the segment is empty)
    int plan_ok;                 // vb_plan's histogram (taken under the PREVIOUS run's geometry, in the pass that takes the bbox) is valid for this run's geometry
};

constexpr int VOX_HT = 1024;     // keys per workgroup in the head (segment boundary) kernels
constexpr int VB_NB = 4096;      // binned path: bins per segment (LDS histogram of the partition kernels)
constexpr int VB_CL_LOG = 10;    // binned path: voxels accumulated in LDS per sweep = 1 << VB_CL_LOG
constexpr int VB_TILE = 4096;    // binned path: points per tile of the histogram kernel (grid-stride)
constexpr int VB_STILE = 4096;   // binned path: points per workgroup of the scatter kernel (2048: 62 us instead of 56 us for the 4.9 M-point map)
constexpr int VB_PAD = 1;        // binned path: stride of the global bin counters / cursors (one per 64-B line, VB_PAD = 16, measured SLOWER: hist 36 vs 29 us, scatter 67 vs 56 us)
constexpr int VB_WG = 512;       // deterministic partition: workgroups (= contiguous point ranges) per segment (256 / 512 / 1024: scatter 62 / 44 / 46 us on the 4.87 M-point map)
constexpr int VB_WROW = VB_NB + 64;   // words per row of the per-(workgroup, bin) count table: NOT a power of two — vb_colscan walks columns, and rows 16 KB apart meet in the same memory channels
constexpr int VB_CH = 8192;       // binned path: points per accumulate workgroup (chunk of a bin); with 32 scans in flight 4096 / 8192 / 16384: 7 050 / 7 400 / 7 460 scans/s (one rebuild alone: accum 33 / ~40 / 50 us)
constexpr int VB_LIGHT = 256;     // binned path: a bin of at most this many points is accumulated by ONE wavefront (vb_light_kernel)
constexpr int VB_TAB = 1 << VB_CL_LOG;      // entries per compacted chunk table
constexpr int VB_TABC = VB_TAB + 32;         // u32 words per chunk table: entries, then the entry count
constexpr int VB_ACC_BLOCKS = 1024;
enum VoxMode { VOX_AUTO = 0, VOX_SORTED = 1, VOX_BINNED = 2 };

struct VoxelPlan {
    int nseg = 0, seg_cap = 0, nblk_h = 0;
    bool concat_out = false;     // outputs of the segments are concatenated in out[0]'s array
    int centroid_lanes = 8;      // lanes that share one output voxel (32 for plans with tens of points per voxel)
    SortPlan sort;
    VoxSegStatic* d_static = nullptr;
    VoxSegDyn* d_dyn = nullptr;
    VoxGrid* d_grid = nullptr;
    int *d_n = nullptr, *d_nbits = nullptr;
    int* d_blockHeads = nullptr;   // [nseg][nblk_h]
    int* d_starts = nullptr;       // [nseg][seg_cap + 1]  first sorted position of every output voxel
    int* d_nout = nullptr;         // [nseg + 1]  voxels per segment, [nseg] = total
    float* d_mmPartial = nullptr;  // [nseg][nblk_mm][12] per-workgroup bbox partials
    int nblk_mm = 0;
    // binned path
    int n_host[4] = {0, 0, 0, 0};     // plans whose segment lengths the host knows (the raw map): passed as kernel arguments
    bool use_n_host = false;          // instead of d_dyn[].n (in_off stays 0)
    const int* n_dev[4] = {nullptr, nullptr, nullptr, nullptr};   // or: device counters of the producer (scan grids), in_off 0
    int mode = VOX_AUTO;
    unsigned* d_binCount = nullptr;   // [nseg][VB_NB]    points per bin (zero between runs)
    int* d_binStart = nullptr;        // [nseg][VB_NB+1]  first bucketed position of every bin
    unsigned* d_cursor = nullptr;     // [nseg][VB_NB]    reservation cursor of the partition
    int* d_binVox = nullptr;          // [nseg][VB_NB]    occupied voxels per bin
    int* d_binOut = nullptr;          // [nseg][VB_NB]    exclusive scan of d_binVox
    lvi_pt* d_bucketed = nullptr;     // [nseg][seg_cap]  points grouped by bin
    lvi_pt* d_staging = nullptr;      // [nseg][seg_cap]  centroids of bin b at d_binStart[b]…
    uint2* d_stagingKC = nullptr;     // [nseg][seg_cap]  (voxel idx, point count) beside them
    int* d_chunkStart = nullptr;      // [nseg][VB_NB+1]  exclusive scan of the chunks per bin
    int* d_multiStart = nullptr;      // [nseg][VB_NB+1]  … of the chunks of bins with more than one chunk
    int* d_chunkBin = nullptr;        // [nseg][max_chunks] bin of every chunk (saves the accumulate kernel a 12-step search through L2)
    int* d_multiOwner = nullptr;      // [nseg][max_multi] first chunk-table slot of a multi-chunk bin -> the bin (else -1)
    int* d_lightBin = nullptr;        // [nseg][VB_NB + 1] bins of at most VB_LIGHT points (one wavefront each, vb_light_kernel); their number at [VB_NB]
    int max_chunks = 0;
    unsigned long long* d_chunkTabV = nullptr;   // [nseg][max_multi][4][VB_TAB] compacted chunk tables of multi-chunk bins (sums of the occupied cells)
    unsigned* d_chunkTabC = nullptr;             // [nseg][max_multi][VB_TABC]: (cell << 16 | count) per entry, entry count at [VB_TAB]
    int max_multi = 0;
    unsigned long long* h_ncells = nullptr;   // pinned host, [nseg]: div_b product of the latest run (AUTO's hint)
    mutable int last_mode = VOX_SORTED;       // what the latest voxel_downsample_batch enqueued
    int bin_pts = 2048, bin_max = 1024;       // binned path: points aimed at per bin, most bins (<= VB_NB)
    unsigned* d_wprefix = nullptr;            // [nseg][VB_WG][VB_NB] per-(workgroup, bin) prefix of the deterministic partition, plans that cache only
    unsigned* d_binCountCached = nullptr;     // [nseg][VB_NB] points per bin of the current input (voxel_bbox_pass), plans that cache only
    mutable bool hist_cached = false;         // … valid (binned realisation was the resolved one when the pass ran)
    bool plan_per_run = false;                // the raw local map in its reference-faithful form: bbox and per-bin counts are taken inside EVERY run (vb_plan: one
                                              // pass for both, the histogram under the previous run's grid geometry, re-taken by vb_hist_w when the geometry moved)
    int* d_planMiss = nullptr;                // [nseg] a point fell outside the previous geometry (vb_plan)
    bool bbox_cached = false;                 // d_mmPartial holds the bbox partials of the CURRENT input (voxel_bbox_pass ran after the input was written)

    template <class AR> void allocate(AR& ar, int nseg_, int seg_cap_, bool concat)
    {
        nseg = nseg_; seg_cap = seg_cap_; concat_out = concat; nblk_h = div_up(seg_cap_, VOX_HT);
        sort.allocate(ar, nseg_, seg_cap_);
        const size_t tot = (size_t)nseg_ * seg_cap_;
        d_static = ar.template alloc<VoxSegStatic>(nseg_);
        d_dyn = ar.template alloc<VoxSegDyn>(nseg_);
        d_grid = ar.template alloc<VoxGrid>(nseg_);
        d_n = ar.template alloc<int>(nseg_);
        d_nbits = ar.template alloc<int>(nseg_);
        d_blockHeads = ar.template alloc<int>((size_t)nseg_ * nblk_h);
        d_starts = ar.template alloc<int>((size_t)nseg_ * ((size_t)seg_cap_ + 1));
        d_nout = ar.template alloc<int>(nseg_ + 1);
        nblk_mm = std::max(1, std::min(div_up(seg_cap_, 256 * 16), 1024));
        d_mmPartial = ar.template alloc<float>((size_t)nseg_ * nblk_mm * 12);
        d_binCount = ar.template alloc<unsigned>((size_t)nseg_ * VB_NB * VB_PAD);
        d_binStart = ar.template alloc<int>((size_t)nseg_ * (VB_NB + 1));
        d_cursor = ar.template alloc<unsigned>((size_t)nseg_ * VB_NB * VB_PAD);
        d_binVox = ar.template alloc<int>((size_t)nseg_ * VB_NB);
        d_binOut = ar.template alloc<int>((size_t)nseg_ * VB_NB);
        d_bucketed = ar.template alloc<lvi_pt>(tot);
        d_staging = ar.template alloc<lvi_pt>(tot);
        d_stagingKC = ar.template alloc<uint2>(tot);
        d_chunkStart = ar.template alloc<int>((size_t)nseg_ * (VB_NB + 1));
        d_multiStart = ar.template alloc<int>((size_t)nseg_ * (VB_NB + 1));
        max_chunks = div_up(seg_cap_, VB_CH) + VB_NB;
        d_chunkBin = ar.template alloc<int>((size_t)nseg_ * max_chunks);
        d_lightBin = ar.template alloc<int>((size_t)nseg_ * (VB_NB + 1));
        max_multi = 2 * div_up(seg_cap_, VB_CH) + 2;         // sum of ceil(cnt/CH) over bins with cnt > CH  <=  2 n / CH
        d_multiOwner = ar.template alloc<int>((size_t)nseg_ * max_multi);
        d_chunkTabV = ar.template alloc<unsigned long long>((size_t)nseg_ * max_multi * (4 * VB_TAB));
        d_chunkTabC = ar.template alloc<unsigned>((size_t)nseg_ * max_multi * VB_TABC);
        d_planMiss = ar.template alloc<int>(nseg_);
    }
    void release();                                                            // frees h_ncells
    void set_static(const Ctx& ctx, const VoxSegStatic* host_segs);           // H2D of the per-segment pointers (+ the pinned hint)
};

// ---------------------------------------------------------------------------------------------------------------------
// Incremental local map: the two VoxelGrids of extractCloud (mapOptimization.cpp:958-965) kept as
// persistent per-voxel sums.  A voxel's membership — floor(p * inv_leaf) per axis — and its sums — relative to the voxel's
// own origin — do not depend on the bounding box, so whole keyframes can be added to and taken from the sums; an emission
// computes PCL's linear idx of every live voxel for the CURRENT bounding box, sorts by it and writes the centroids: the same
// voxels, order and bits as filtering the fused cloud.  One open-addressing table per map kind (corner, surf).
struct IncPiece { int in_off, n, which, sign, kf; float A[12]; };      // one (keyframe, corner|surf) cloud to add (+1) or take out (-1)
enum { INC_ERR_RANGE = 1, INC_ERR_FULL = 2, INC_ERR_OVERFLOW = 4, INC_ERR_INTENSITY = 8 };
struct IncMap {
    int H = 0;                                // slots per table (power of two)
    int max_kf = 0;
    unsigned long long* key[2] = {nullptr, nullptr};       // [H] packed absolute voxel coordinates, ~0 = empty
    unsigned long long* sums[2] = {nullptr, nullptr};      // [H][4]
    int* cnt[2] = {nullptr, nullptr};                      // [H] points in the voxel (0: a voxel every keyframe has left)
    int* occ[2] = {nullptr, nullptr};                      // [H] slots in first-touch order
    int* d_nocc = nullptr;                                 // [2]
    unsigned* kfBox = nullptr;                             // [max_kf][2][8] map-frame bbox of a stored keyframe at the pose it was added with
    int* d_active = nullptr;                               // [max_active] key indices of the current list (bbox fold)
    int* h_active = nullptr;                               // pinned staging of the same
    int max_active = 0;
    IncPiece* d_pieces = nullptr; IncPiece* h_pieces = nullptr; int max_pieces = 0;
    SortPlan sort;                                         // (idx, slot) pairs of the live voxels, 2 segments
    int *d_n = nullptr, *d_nbits = nullptr;                // [2]
    int* d_status = nullptr;                               // INC_ERR_* (sticky until cleared)
    int* h_status = nullptr;                               // pinned: [0] status, [1..2] n_occ
    template <class AR> void allocate(AR& ar, int H_, int max_kf_, int max_active_)
    {
        H = H_; max_kf = max_kf_; max_active = max_active_; max_pieces = 4 * max_active_ + 16;
        for (int w = 0; w < 2; w++) {
            key[w] = ar.template alloc<unsigned long long>(H); sums[w] = ar.template alloc<unsigned long long>((size_t)H * 4);
            cnt[w] = ar.template alloc<int>(H); occ[w] = ar.template alloc<int>(H);
        }
        d_nocc = ar.template alloc<int>(2);
        kfBox = ar.template alloc<unsigned>((size_t)std::max(max_kf, 1) * 16);
        d_active = ar.template alloc<int>(std::max(max_active, 1));
        d_pieces = ar.template alloc<IncPiece>(max_pieces);
        sort.allocate(ar, 2, H / 2);
        d_n = ar.template alloc<int>(2); d_nbits = ar.template alloc<int>(2);
        d_status = ar.template alloc<int>(4);
    }
};
void incmap_clear(const Ctx& ctx, const IncMap& m);
// add / remove the listed keyframe clouds (pool = the keyframe store)
void incmap_apply(const Ctx& ctx, const IncMap& m, const lvi_pt* pool, int n_pieces, int max_n, const float leaf[2]);
// emit the live voxels of both tables as laserCloud{Corner,Surf}FromMapDS: fills grid[2] / nout[3] / out[2] exactly as
// voxel_downsample_batch of the fused cloud would (n_active unique keys of d_active give the bounding box)
void incmap_emit(const Ctx& ctx, const IncMap& m, int n_active, const float leaf[2], VoxGrid* grid, int* nout, lvi_pt* outC, lvi_pt* outS, int seg_cap);

// Enqueue the whole filter for every segment.  Two interchangeable realisations with bit-identical output:
//   SORTED  stable radix sort of (voxel idx, point index) + ordered head compaction + per-voxel sums; any grid.
//   BINNED  one partition of the points into <= VB_NB bins of consecutive voxel idx, then one workgroup per bin
//           accumulates its voxels in LDS; ~2.5x less HBM traffic and 8 launches instead of 19 when the grid is
//           compact (div_b product <= VB_NB << VB_CL_LOG); still correct, but sweeping each bin several times,
//           when it is not.
// AUTO enqueues BINNED when the previous run's grids (read from pinned host memory, no sync) had at most
// 16.8 M cells (four sweeps per bin), SORTED for sparser ones.  d_dyn must have been written (on the same
// stream) by the producer.  n_hint: nominal total input points, for byte accounting only.
void voxel_downsample_batch(const Ctx& ctx, const VoxelPlan& plan, const char* tag, double n_hint);
// the same plan of S batch slots (identical shapes) in ONE launch sequence, blockIdx.z = slot; n_hint = points of all slots
void voxel_downsample_batch(const Ctx& ctx, const VoxelPlan* const* plans, int S, const char* tag, double n_hint);
// the bbox partial records of the plan's current input, as a pass of its own (sets nothing on the plan: the caller owns
// bbox_cached and clears it whenever the input changes)
void voxel_bbox_pass(const Ctx& ctx, const VoxelPlan& plan, const char* tag, double n_hint);
// the realisation the next voxel_downsample_batch of this plan will enqueue (AUTO resolved from the previous batch's hint)
int voxel_resolve_mode(const VoxelPlan& plan);
// a cloud of at most VOX_TINY points, one workgroup, one launch, pinned host memory in and out (the node's key-pose grid)
constexpr int VOX_TINY = 1024;
void voxel_tiny(const Ctx& ctx, const lvi_pt* in_pinned, int n, float leaf, int seg_cap, int bin_pts, int bin_max, lvi_pt* out_pinned, int* hdr_pinned,
                int* cells_pinned, int* counts_pinned, int* keys_pinned);

}  // namespace lvi
