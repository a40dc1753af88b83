// Deferred partial-sum folds (see vg_colsum_f32_multi_kernel in norm.hip).
#pragma once
#define VG_MAX_FOLD_JOBS 42
struct VgFoldJob {
  const float* part; int rows, width;   // partial rows [rows][width]
  float* dst[4]; int n[4];              // up to 4 consecutive column segments, accumulated (+=) into dst (nullptr = skip)
};
struct VgFoldJobs { int n; VgFoldJob j[VG_MAX_FOLD_JOBS]; };
// returns -1 (nothing queued) when the queue is full: the layouts reject depths that could get there (vg_vit_layout /
// vg_gen_layout), this is the second line of defence against writing past the array
static inline int vg_fold_push(VgFoldJobs& q, const float* part, int rows, int width, float* d0, int n0, float* d1, int n1,
                               float* d2, int n2, float* d3, int n3) {
  if (q.n >= VG_MAX_FOLD_JOBS) return -1;
  VgFoldJob& J = q.j[q.n++];
  J.part = part; J.rows = rows; J.width = width;
  J.dst[0] = d0; J.n[0] = n0; J.dst[1] = d1; J.n[1] = n1; J.dst[2] = d2; J.n[2] = n2; J.dst[3] = d3; J.n[3] = n3;
  return 0;
}
