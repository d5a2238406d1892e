// placeholder until the tracker kernels land (next commit): the symbols exist, compute fails loudly
#include "lvi_dev.hpp"
namespace { int32_t nyi() { lvi::set_error("tracker HIP path not built yet"); return LVI_ERR_UNSUPPORTED; } }
extern "C" {
void lvi_tracker_params_default(lvi_tracker_params* p)
{
    memset(p, 0, sizeof(*p));
    p->max_width = 1280; p->max_height = 720; p->max_cnt = 150; p->min_dist = 20.0;
    p->lk_win = 21; p->lk_max_level = 3; p->lk_max_iters = 30; p->lk_eps = 0.01; p->lk_min_eig_threshold = 1e-4f;
    p->gftt_quality = 0.01; p->max_features = 1024;
}
int32_t lvi_tracker_create(const lvi_tracker_params*, int32_t, lvi_tracker**) { return nyi(); }
void lvi_tracker_destroy(lvi_tracker*) {}
int32_t lvi_tracker_sync(lvi_tracker*) { return nyi(); }
int32_t lvi_lk_track(lvi_tracker*, const uint8_t*, const uint8_t*, int32_t, int32_t, int32_t, const float*, int32_t, float*, uint8_t*, float*) { return nyi(); }
int32_t lvi_good_features(lvi_tracker*, const uint8_t*, const uint8_t*, int32_t, int32_t, int32_t, int32_t, double, double, float*, int32_t, int32_t*) { return nyi(); }
int32_t lvi_tracker_push_image(lvi_tracker*, const uint8_t*, int32_t, int32_t, int32_t) { return nyi(); }
int32_t lvi_tracker_set_points(lvi_tracker*, const float*, int32_t) { return nyi(); }
int32_t lvi_tracker_run_lk(lvi_tracker*) { return nyi(); }
int32_t lvi_tracker_get_lk(lvi_tracker*, float*, uint8_t*, float*, int32_t, int32_t*) { return nyi(); }
int32_t lvi_tracker_set_mask(lvi_tracker*, const uint8_t*, int32_t, int32_t, int32_t) { return nyi(); }
int32_t lvi_tracker_run_gftt(lvi_tracker*, int32_t) { return nyi(); }
int32_t lvi_tracker_get_gftt(lvi_tracker*, float*, int32_t, int32_t*) { return nyi(); }
int32_t lvi_tracker_debug_get(lvi_tracker*, int32_t, void*, int64_t, int64_t*) { return nyi(); }
int32_t lvi_tracker_prof_enable(lvi_tracker*, int32_t) { return nyi(); }
int32_t lvi_tracker_prof_reset(lvi_tracker*) { return nyi(); }
int32_t lvi_tracker_prof_read(lvi_tracker*, lvi_kernel_stat*, int32_t, int32_t*) { return nyi(); }
}
