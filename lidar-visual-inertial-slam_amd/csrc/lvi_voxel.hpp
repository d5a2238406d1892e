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
};

constexpr int VOX_SMALL_MAX = 4096;     // only tiny plans take the single-workgroup path: measured on MI355X, one CU is
                                        // latency-bound beyond a few thousand points (33 k points: 0.5 ms vs 0.15 ms multi-workgroup)
constexpr int VOX_HT = 1024;     // keys per workgroup in the head (segment boundary) kernels

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
    float* d_mmPartial = nullptr;  // [nseg][nblk_mm][8] per-workgroup bbox partials
    int nblk_mm = 0;

    template <class AR> void allocate(AR& ar, int nseg_, int seg_cap_, bool concat)
    {
        nseg = nseg_; seg_cap = seg_cap_; concat_out = concat; nblk_h = div_up(seg_cap_, VOX_HT);
        sort.allocate(ar, nseg_, seg_cap_);
        d_static = ar.template alloc<VoxSegStatic>(nseg_);
        d_dyn = ar.template alloc<VoxSegDyn>(nseg_);
        d_grid = ar.template alloc<VoxGrid>(nseg_);
        d_n = ar.template alloc<int>(nseg_);
        d_nbits = ar.template alloc<int>(nseg_);
        d_blockHeads = ar.template alloc<int>((size_t)nseg_ * nblk_h);
        d_starts = ar.template alloc<int>((size_t)nseg_ * ((size_t)seg_cap_ + 1));
        d_nout = ar.template alloc<int>(nseg_ + 1);
        nblk_mm = std::max(1, std::min(div_up(seg_cap_, 256 * 16), 1024));
        d_mmPartial = ar.template alloc<float>((size_t)nseg_ * nblk_mm * 8);
    }
    void set_static(const Ctx& ctx, const VoxSegStatic* host_segs) const;     // H2D of the per-segment pointers
};

// Enqueue the whole filter for every segment.  d_dyn must have been written (on the same
// stream) by the producer.  n_hint: nominal total input points, for byte accounting only.
void voxel_downsample_batch(const Ctx& ctx, const VoxelPlan& plan, const char* tag, double n_hint);

}  // namespace lvi
