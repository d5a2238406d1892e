// Stable LSD radix sort of (u32 key, u32 value) pairs, batched over independent segments
// whose lengths and key widths live in device memory (no host sync anywhere).
#pragma once
#include "lvi_dev.hpp"

namespace lvi {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;      // 4096 pairs per workgroup
constexpr int RS_ITEMS_SMALL = 4;                   // 1024-pair tiles: plans whose live size is a few 10^4 pairs whatever the capacity (GFTT candidates)

struct SortPlan {
    int nseg = 0;
    int seg_cap = 0;                 // every segment owns [s*seg_cap, (s+1)*seg_cap) of the four arrays
    int nblk = 0;                    // tiles per segment = ceil(seg_cap / (RS_THREADS * items))
    int items = RS_ITEMS;            // pairs per thread and tile: RS_ITEMS or RS_ITEMS_SMALL
    unsigned *keysA = nullptr, *valsA = nullptr, *keysB = nullptr, *valsB = nullptr;
    unsigned* hist = nullptr;        // [nseg][256][nblk]  per-tile digit counts, scanned in place
    unsigned* digitTotal = nullptr;  // [nseg][256]

    template <class AR> void allocate(AR& ar, int nseg_, int seg_cap_, int items_ = RS_ITEMS)
    {
        nseg = nseg_; seg_cap = seg_cap_; items = items_; nblk = div_up(seg_cap_, RS_THREADS * items_);
        size_t tot = (size_t)nseg * seg_cap;
        keysA = ar.template alloc<unsigned>(tot); valsA = ar.template alloc<unsigned>(tot);
        keysB = ar.template alloc<unsigned>(tot); valsB = ar.template alloc<unsigned>(tot);
        hist = ar.template alloc<unsigned>((size_t)nseg * 256 * nblk);
        digitTotal = ar.template alloc<unsigned>((size_t)nseg * 256);
    }
};

// Sort every segment s in place of keysA/valsA (input) over its first d_n[s] entries, by the low
// d_nbits[s] bits of the key.  max_passes bounds the passes launched (ceil(max key bits / 8)).
// The sorted result of segment s is in (keysA,valsA) when ceil(d_nbits[s]/8) is even, else in
// (keysB,valsB) — use rs_result_in_B() on the device.
// n_hint = expected total number of pairs (host's nominal figure, used only for the profiler's
// algorithmic-byte accounting).
void radix_sort_pairs(const Ctx& ctx, const SortPlan& plan, const int* d_n, const int* d_nbits, int max_passes, const char* tag, double n_hint);

#ifdef __HIPCC__
__device__ __forceinline__ bool rs_result_in_B(int nbits) { return (((nbits + 7) >> 3) & 1) != 0; }
#endif

}  // namespace lvi
