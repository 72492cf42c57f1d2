// Leaf-selection statistics (see lg_leaf.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/leafgrasp.h"

struct LgLeafWs;
// per-kernel event timing of the stage (lg_profile_enable / lg_profile_read: "leaf_presence", "leaf_accumulate", "leaf_hist",
// "leaf_select", "leaf_edt", "leaf_pack"); owned by the handle, passed to the run functions (NULL: no timing)
struct LgLeafProf;
LgLeafProf* lg_leaf_prof_new();
void lg_leaf_prof_free(LgLeafProf* p);
void lg_leaf_prof_enable(LgLeafProf* p, int on);
int lg_leaf_prof_read(LgLeafProf* p, const char* name, int* launches, double* total_ms);   // 1 if the name is one of the stage's
int lg_leaf_run(LgLeafWs*& w, const int16_t* labels, const float* depth, int H, int W, float cx, float cy, float f,
                lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, hipStream_t s, hipStream_t side,
                std::string* err, LgLeafProf* prof = nullptr);
// B frames [B][H][W]; stats [B][max_leaves], n_leaves [B], extrema [B][4], status [B] (per-frame lg_status; NULL: the
// first failing frame's status is returned).  `side`: a second stream of the same handle for the clutter-extrema chain
// (NULL: everything on s)
int lg_leaf_run_batch(LgLeafWs*& w, const int16_t* labels, const float* depth, int B, int H, int W, float cx, float cy,
                      float f, lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, int* status,
                      hipStream_t s, hipStream_t side, std::string* err, LgLeafProf* prof = nullptr);
// statistics + the selection of leaf_scorer.py:53-181 for B frames in one call: ids [B] (-1: none; -2: a frame the caller
// should take through the general path -- more than 256 labels or 128 leaves), n_tall [B], tall [B][tall_cap]
// the selection of leaf_scorer.py:53-203 from one frame's statistics rows (host only); returns the label, -1 (none) or -2 (>= 128 leaves)
int lg_leaf_select_host(const lg_leaf_stat* st, int n, const int32_t ext[4], int H, int W, double cx, double cy, double f,
                        int32_t* tall, int tall_cap, int* n_tall);
int lg_leaf_select_batch_run(LgLeafWs*& w, const int16_t* labels, const float* depth, int B, int H, int W, double cx, double cy,
                             double f, int32_t* ids, int32_t* n_tall, int32_t* tall, int tall_cap, hipStream_t s, hipStream_t side,
                             std::string* err, LgLeafProf* prof = nullptr);
void lg_leaf_free(LgLeafWs*& w);
