// Leaf-selection statistics (see lg_leaf.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/leafgrasp.h"

struct LgLeafWs;
int lg_leaf_run(LgLeafWs*& w, const int16_t* labels, const float* depth, int H, int W, float cx, float cy, float f,
                lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, hipStream_t s, std::string* err);
void lg_leaf_free(LgLeafWs*& w);
