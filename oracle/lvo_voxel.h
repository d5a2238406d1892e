// TEST INFRASTRUCTURE — CPU oracle.  PARITY UNPINNED (PCL source not in the reference
// tree; see lvo_math.h).
//
// pcl::VoxelGrid<pcl::PointXYZI>::applyFilter (PCL 1.12 filters/impl/voxel_grid.hpp) with
// the defaults the reference leaves in place: downsample_all_data_=true,
// min_points_per_voxel_=0, no filter field, dense input.  Call sites:
// featureExtraction.cpp:61,240-241; mapOptimization.cpp:247-250, 959-964, 991-997.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../include/lvi_hotpath.h"

namespace lvo {

struct VoxelDebug {
    std::vector<int32_t> keys;     // per input point
    std::vector<int32_t> cells;    // distinct keys ascending
    std::vector<int32_t> counts;   // points per cell
    bool overflow = false;
};

struct cloud_point_index_idx {
    unsigned int idx;
    unsigned int cloud_point_index;
    bool operator<(const cloud_point_index_idx& p) const { return idx < p.idx; }
};

// returns number of output points; out must hold n points (overflow rule returns the input unchanged)
inline int voxel_grid_filter(const lvi_pt* in, int n, float leaf, std::vector<lvi_pt>& out, VoxelDebug* dbg = nullptr)
{
    out.clear();
    if (dbg) { dbg->keys.clear(); dbg->cells.clear(); dbg->counts.clear(); dbg->overflow = false; }
    if (n <= 0) return 0;
    // setLeafSize: inverse_leaf_size_ = 1 / leaf_size_ (Eigen Array4f)
    const float inv = 1.0f / leaf;
    // getMinMax3D (dense cloud)
    float min_p[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float max_p[3] = {-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(), -std::numeric_limits<float>::max()};
    for (int i = 0; i < n; i++) {
        const float v[3] = {in[i].x, in[i].y, in[i].z};
        for (int d = 0; d < 3; d++) { min_p[d] = std::min(min_p[d], v[d]); max_p[d] = std::max(max_p[d], v[d]); }
    }
    // overflow rule
    std::int64_t dx = static_cast<std::int64_t>((max_p[0] - min_p[0]) * inv) + 1;
    std::int64_t dy = static_cast<std::int64_t>((max_p[1] - min_p[1]) * inv) + 1;
    std::int64_t dz = static_cast<std::int64_t>((max_p[2] - min_p[2]) * inv) + 1;
    if ((dx * dy * dz) > static_cast<std::int64_t>(std::numeric_limits<std::int32_t>::max())) {
        out.assign(in, in + n);            // "output = *input_" — no downsampling
        if (dbg) dbg->overflow = true;
        return n;
    }
    int min_b[3], max_b[3], div_b[3], divb_mul[3];
    for (int d = 0; d < 3; d++) {
        min_b[d] = static_cast<int>(std::floor(min_p[d] * inv));
        max_b[d] = static_cast<int>(std::floor(max_p[d] * inv));
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    divb_mul[0] = 1; divb_mul[1] = div_b[0]; divb_mul[2] = div_b[0] * div_b[1];

    std::vector<cloud_point_index_idx> index_vector;
    index_vector.reserve(n);
    for (int i = 0; i < n; i++) {
        int ijk0 = static_cast<int>(std::floor(in[i].x * inv) - static_cast<float>(min_b[0]));
        int ijk1 = static_cast<int>(std::floor(in[i].y * inv) - static_cast<float>(min_b[1]));
        int ijk2 = static_cast<int>(std::floor(in[i].z * inv) - static_cast<float>(min_b[2]));
        int idx = ijk0 * divb_mul[0] + ijk1 * divb_mul[1] + ijk2 * divb_mul[2];
        index_vector.push_back({static_cast<unsigned int>(idx), static_cast<unsigned int>(i)});
    }
    if (dbg) { dbg->keys.resize(n); for (int i = 0; i < n; i++) dbg->keys[i] = (int32_t)index_vector[i].idx; }
    // unstable std::sort by idx only — same libstdc++ introsort as the reference platform
    std::sort(index_vector.begin(), index_vector.end(), std::less<cloud_point_index_idx>());

    size_t index = 0;
    while (index < index_vector.size()) {
        size_t i = index + 1;
        while (i < index_vector.size() && index_vector[i].idx == index_vector[index].idx) ++i;
        // CentroidPoint<PointXYZI>: f32 running sums of xyz and intensity, divided by the count
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (size_t li = index; li < i; ++li) {
            const lvi_pt& p = in[index_vector[li].cloud_point_index];
            sx += p.x; sy += p.y; sz += p.z; si += p.intensity;
        }
        const float cnt = static_cast<float>(i - index);
        lvi_pt o; o.x = sx / cnt; o.y = sy / cnt; o.z = sz / cnt; o.intensity = si / cnt;
        out.push_back(o);
        if (dbg) { dbg->cells.push_back((int32_t)index_vector[index].idx); dbg->counts.push_back((int32_t)(i - index)); }
        index = i;
    }
    return (int)out.size();
}

}  // namespace lvo
