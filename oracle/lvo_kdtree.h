// TEST INFRASTRUCTURE — CPU oracle.  See lvo_math.h for the provenance note
// (PARITY UNPINNED: PCL/FLANN sources are not in the reference tree).
//
// pcl::KdTreeFLANN<PointXYZI>::setInputCloud + nearestKSearch(p, 5, idx, sqd)
// as called at mapOptimization.cpp:1322-1323, 1019, 1111.
// PCL 1.12 builds flann::Index<L2_Simple<float>>(points xyz, KDTreeSingleIndexParams(15))
// and queries with SearchParams(-1, epsilon=0): exact k-NN, ascending squared L2.
// This file restates FLANN 1.9 KDTreeSingleIndex (kdtree_single_index.h): bounding-box
// mid-split build with reordered points, leaf size 15, branch-and-bound search,
// KNNSimpleResultSet.  Distance = ((dx*dx)+dy*dy)+dz*dz in f32, no FMA.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <vector>

namespace lvo {

class KdTree3f {
public:
    struct Interval { float low, high; };
    struct Node {
        int left, right;        // leaf: point range [left,right)
        int divfeat;
        float divlow, divhigh;
        int child1, child2;     // -1 for leaf
    };

    void build(const float* xyz, int stride_floats, int n)
    {
        n_ = n;
        nodes_.clear(); vind_.resize(n); pts_.resize((size_t)n * 3);
        src_.resize((size_t)n * 3);
        for (int i = 0; i < n; i++) {
            vind_[i] = i;
            src_[3 * (size_t)i + 0] = xyz[(size_t)i * stride_floats + 0];
            src_[3 * (size_t)i + 1] = xyz[(size_t)i * stride_floats + 1];
            src_[3 * (size_t)i + 2] = xyz[(size_t)i * stride_floats + 2];
        }
        if (n == 0) { root_ = -1; return; }
        // computeBoundingBox
        for (int d = 0; d < 3; d++) { root_bbox_[d].low = src_[d]; root_bbox_[d].high = src_[d]; }
        for (int k = 1; k < n; k++)
            for (int d = 0; d < 3; d++) {
                float v = src_[3 * (size_t)k + d];
                if (v < root_bbox_[d].low) root_bbox_[d].low = v;
                if (v > root_bbox_[d].high) root_bbox_[d].high = v;
            }
        nodes_.reserve((size_t)n / 4 + 16);
        Interval bbox[3] = {root_bbox_[0], root_bbox_[1], root_bbox_[2]};
        root_ = divideTree(0, n, bbox);
        root_bbox_[0] = bbox[0]; root_bbox_[1] = bbox[1]; root_bbox_[2] = bbox[2];
        // reorder_ = true
        for (int i = 0; i < n; i++)
            for (int d = 0; d < 3; d++) pts_[3 * (size_t)i + d] = src_[3 * (size_t)vind_[i] + d];
    }

    int size() const { return n_; }

    // exact k-NN, k <= 8.  Returns number found (min(k, n)); idx/sqd ascending.
    int knn(const float q[3], int k, int* idx, float* sqd) const
    {
        float rd[8]; int ri[8]; int cnt = 0;
        for (int i = 0; i < k; i++) { rd[i] = FLT_MAX; ri[i] = -1; }
        if (root_ < 0) return 0;
        float dists[3] = {0.f, 0.f, 0.f};
        float distsq = 0.f;
        for (int d = 0; d < 3; d++) {       // computeInitialDistances
            if (q[d] < root_bbox_[d].low)  { float t = q[d] - root_bbox_[d].low;  dists[d] = t * t; distsq += dists[d]; }
            if (q[d] > root_bbox_[d].high) { float t = q[d] - root_bbox_[d].high; dists[d] = t * t; distsq += dists[d]; }
        }
        searchLevel(q, root_, distsq, dists, k, rd, ri, cnt);
        for (int i = 0; i < cnt; i++) { idx[i] = ri[i]; sqd[i] = rd[i]; }
        return cnt;
    }

private:
    int n_ = 0, root_ = -1;
    std::vector<Node> nodes_;
    std::vector<int> vind_;
    std::vector<float> pts_, src_;
    Interval root_bbox_[3];
    static constexpr int kLeafMax = 15;
    static constexpr float kEPS = 0.00001f;

    int divideTree(int left, int right, Interval bbox[3])
    {
        int me = (int)nodes_.size();
        nodes_.push_back(Node());
        if ((right - left) <= kLeafMax) {
            nodes_[me].child1 = nodes_[me].child2 = -1;
            nodes_[me].left = left; nodes_[me].right = right;
            for (int d = 0; d < 3; d++) { bbox[d].low = src_[3 * (size_t)vind_[left] + d]; bbox[d].high = bbox[d].low; }
            for (int k = left + 1; k < right; k++)
                for (int d = 0; d < 3; d++) {
                    float v = src_[3 * (size_t)vind_[k] + d];
                    if (bbox[d].low > v) bbox[d].low = v;
                    if (bbox[d].high < v) bbox[d].high = v;
                }
        } else {
            int idx, cutfeat; float cutval;
            middleSplit(&vind_[left], right - left, idx, cutfeat, cutval, bbox);
            nodes_[me].divfeat = cutfeat;
            Interval lb[3] = {bbox[0], bbox[1], bbox[2]};
            lb[cutfeat].high = cutval;
            int c1 = divideTree(left, left + idx, lb);
            Interval rb[3] = {bbox[0], bbox[1], bbox[2]};
            rb[cutfeat].low = cutval;
            int c2 = divideTree(left + idx, right, rb);
            nodes_[me].child1 = c1; nodes_[me].child2 = c2;
            nodes_[me].divlow = lb[cutfeat].high;
            nodes_[me].divhigh = rb[cutfeat].low;
            for (int d = 0; d < 3; d++) {
                bbox[d].low = std::min(lb[d].low, rb[d].low);
                bbox[d].high = std::max(lb[d].high, rb[d].high);
            }
        }
        return me;
    }

    void computeMinMax(const int* ind, int count, int dim, float& mn, float& mx) const
    {
        mn = src_[3 * (size_t)ind[0] + dim]; mx = mn;
        for (int i = 1; i < count; i++) {
            float v = src_[3 * (size_t)ind[i] + dim];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    }

    void middleSplit(int* ind, int count, int& index, int& cutfeat, float& cutval, const Interval bbox[3])
    {
        float max_span = bbox[0].high - bbox[0].low;
        cutfeat = 0;
        cutval = (bbox[0].high + bbox[0].low) / 2;
        for (int i = 1; i < 3; i++) {
            float span = bbox[i].high - bbox[i].low;
            if (span > max_span) { max_span = span; cutfeat = i; cutval = (bbox[i].high + bbox[i].low) / 2; }
        }
        // among near-maximal box spans pick the largest data spread
        float min_elem, max_elem;
        computeMinMax(ind, count, cutfeat, min_elem, max_elem);
        cutval = (min_elem + max_elem) / 2;
        max_span = max_elem - min_elem;
        int k = cutfeat;
        for (int i = 0; i < 3; i++) {
            if (i == k) continue;
            float span = bbox[i].high - bbox[i].low;
            if (span > max_span) {
                computeMinMax(ind, count, i, min_elem, max_elem);
                span = max_elem - min_elem;
                if (span > max_span) { max_span = span; cutfeat = i; cutval = (min_elem + max_elem) / 2; }
            }
        }
        int lim1, lim2;
        planeSplit(ind, count, cutfeat, cutval, lim1, lim2);
        if (lim1 > count / 2) index = lim1;
        else if (lim2 < count / 2) index = lim2;
        else index = count / 2;
        (void)kEPS;
    }

    void planeSplit(int* ind, int count, int cutfeat, float cutval, int& lim1, int& lim2)
    {
        int left = 0, right = count - 1;
        for (;;) {
            while (left <= right && src_[3 * (size_t)ind[left] + cutfeat] < cutval) ++left;
            while (left <= right && src_[3 * (size_t)ind[right] + cutfeat] >= cutval) --right;
            if (left > right) break;
            std::swap(ind[left], ind[right]); ++left; --right;
        }
        lim1 = left;
        right = count - 1;
        for (;;) {
            while (left <= right && src_[3 * (size_t)ind[left] + cutfeat] <= cutval) ++left;
            while (left <= right && src_[3 * (size_t)ind[right] + cutfeat] > cutval) --right;
            if (left > right) break;
            std::swap(ind[left], ind[right]); ++left; --right;
        }
        lim2 = left;
    }

    static inline void addPoint(float dist, int index, int k, float* rd, int* ri, int& cnt)
    {
        // KNNSimpleResultSet::addPoint
        if (dist >= rd[k - 1]) return;
        int i;
        for (i = cnt; i > 0; --i) {
            if (rd[i - 1] > dist) {
                if (i < k) { rd[i] = rd[i - 1]; ri[i] = ri[i - 1]; }
            } else break;
        }
        if (cnt < k) ++cnt;
        rd[i] = dist; ri[i] = index;
    }

    void searchLevel(const float q[3], int ni, float mindistsq, float dists[3], int k, float* rd, int* ri, int& cnt) const
    {
        const Node& node = nodes_[ni];
        if (node.child1 < 0) {
            float worst = rd[k - 1];
            for (int i = node.left; i < node.right; ++i) {
                const float* p = &pts_[3 * (size_t)i];
                float d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
                float dist = 0.f;
                dist += d0 * d0; dist += d1 * d1; dist += d2 * d2;
                if (dist < worst) { addPoint(dist, vind_[i], k, rd, ri, cnt); worst = rd[k - 1]; }
            }
            return;
        }
        int idx = node.divfeat;
        float val = q[idx];
        float diff1 = val - node.divlow;
        float diff2 = val - node.divhigh;
        int best, other; float cut_dist;
        if ((diff1 + diff2) < 0) { best = node.child1; other = node.child2; float t = val - node.divhigh; cut_dist = t * t; }
        else { best = node.child2; other = node.child1; float t = val - node.divlow; cut_dist = t * t; }
        searchLevel(q, best, mindistsq, dists, k, rd, ri, cnt);
        float dst = dists[idx];
        mindistsq = mindistsq + cut_dist - dst;
        dists[idx] = cut_dist;
        if (mindistsq * 1.0f <= rd[k - 1]) searchLevel(q, other, mindistsq, dists, k, rd, ri, cnt);
        dists[idx] = dst;
    }
};

}  // namespace lvo
