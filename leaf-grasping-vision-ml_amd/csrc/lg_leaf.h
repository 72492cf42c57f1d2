// Leaf-selection statistics (see lg_leaf.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/leafgrasp.h"

struct LgLeafWs;
int lg_leaf_run(LgLeafWs*& w, const int16_t* labels, const float* depth, int H, int W, float cx, float cy, float f,
                lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, hipStream_t s, hipStream_t side,
                std::string* err);
// B frames [B][H][W]; stats [B][max_leaves], n_leaves [B], extrema [B][4], status [B] (per-frame lg_status; NULL: the
// first failing frame's status is returned).  `side`: a second stream of the same handle for the clutter-extrema chain
// (NULL: everything on s)
int lg_leaf_run_batch(LgLeafWs*& w, const int16_t* labels, const float* depth, int B, int H, int W, float cx, float cy,
                      float f, lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, int* status,
                      hipStream_t s, hipStream_t side, std::string* err);
void lg_leaf_free(LgLeafWs*& w);
