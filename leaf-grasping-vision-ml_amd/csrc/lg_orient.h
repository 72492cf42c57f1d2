// Device-side leaf orientation (lg_orient.hip): scratch and launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "lg_internal.h"

struct LgOrientWs {
    int capB = 0, H = 0, cap = 0;   // frames, image rows, runs per frame the scratch holds
    size_t lds = 0;
    uint32_t* run_x = nullptr;
    uint16_t* run_y = nullptr;
    int *roots = nullptr, *row_start = nullptr;
    short *rowL = nullptr, *rowR = nullptr;
    unsigned char* vflag = nullptr;
    int* comp = nullptr;            // [B][4][cap] per-component interior pixel count and bounding box (candidate test)
    double* out = nullptr;          // [B][5] theta, long side, short side, centre x, centre y
    int* status = nullptr;          // [B] 0 done on the device, 1 frame needs the host analysis
    double* h_out = nullptr;        // pinned copies
    int* h_status = nullptr;
};

int lg_orient_ensure(LgOrientWs*& w, int B, int H, std::string* err);
void lg_orient_free(LgOrientWs*& w);
// frames [off, off + n) of the scratch; bits / win / fp point at frame `off`.  Writes fp[i] (sin, cos, theta, has_angle),
// out and status; a frame with status 1 has fp[i].has_angle = 0 and must be analysed on the host.
void lg_launch_orient(LgOrientWs* w, const unsigned long long* bits, const LgWin* win, LgFrameParams* fp, int off, int n, int H,
                      int W, int WW, hipStream_t s);
