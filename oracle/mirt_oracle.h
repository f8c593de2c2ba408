/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's per-pixel trace path over the flattened
 * scene arrays of include/mirt.h.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this; the product never does. */
#ifndef MIRT_ORACLE_H
#define MIRT_ORACLE_H
#include "../include/mirt.h"
#ifdef __cplusplus
extern "C" {
#endif

/* visits[0..2] += box tests, leaf visits, primitive tests (may be NULL) */
int mirt_oracle_trace_rays(const mirt_scene_desc* d, uint64_t n, const float* origins, const float* dirs,
                           int32_t* prim_id_out, float* dist_out, float* normal_out, uint64_t* visits);
int mirt_oracle_primary_rays(const mirt_scene_desc* d, uint32_t w, uint32_t h, float* origins, float* dirs);
/* Full image (tile selection in params is honoured: unselected pixels are left untouched).
 * visits9 (may be NULL): box tests, leaf visits, primitive tests for primary / shadow / reflection rays. */
int mirt_oracle_render(const mirt_scene_desc* d, const mirt_render_params* p, uint8_t* bgr, mirt_stats* stats, uint64_t* visits9);
/* every row_step-th row of [row0,row1), every col_step-th column: the bounded sample used for CPU timing */
int mirt_oracle_render_rows(const mirt_scene_desc* d, const mirt_render_params* p, uint32_t row0, uint32_t row1, uint32_t row_step,
                            uint32_t col_step, uint8_t* bgr, mirt_stats* stats, uint64_t* visits9);
/* single primitive / single box, for known-answer tests */
int mirt_oracle_triangle(const float* v9, uint64_t n, const float* origins, const float* dirs, float* dist, float* normal);
int mirt_oracle_sphere(const mirt_sphere* s, uint64_t n, const float* origins, const float* dirs, float* dist, float* normal);
int mirt_oracle_box(const float* bmin, const float* bmax, uint64_t n, const float* origins, const float* dirs, uint8_t* hit);

#ifdef __cplusplus
}
#endif
#endif
