// Device helpers of the HexPlane kernels shared by hexplane.hip and deform_field.hip: ATen's grid_sampler_2d coordinate
// arithmetic (align_corners=True, padding_mode='border'), the kernel argument block and its host-side fill.
#pragma once
#include "mom_common.h"

namespace {

__device__ __forceinline__ float unnorm_clip(float c, int size, float& gmul)
{
    // align_corners=True: ((c+1)/2)*(size-1); border: clip to [0, size-1] with zero gradient when clipped
    float v = ((c + 1.f) / 2.f) * (float)(size - 1);
    gmul = (float)(size - 1) / 2.f;
    if (v <= 0.f) {
        v = 0.f;
        gmul = 0.f;
    } else {
        const float mx = (float)(size - 1);
        if (v >= mx) {
            v = mx;
            gmul = 0.f;
        }
    }
    return v;
}

struct HexArgs {
    int P, levels;
    int res[4][4];
    const float* planes[4][6];
    float* grads[4][6];
    float a0[3], a1[3];  // aabb rows exactly as the reference stores them (row 0 = xyz_max, row 1 = xyz_min)
    float time;
    const float* times;  // optional per-point timestamps [P]; null -> `time` for every point
    const uint32_t* order;  // optional processing order (a permutation of 0..P-1, e.g. Morton order); null -> identity
};

// The per-frame table of time lines (deform_field.hip, hexplane_lines_kernel): for level l and axis k the rows
// lines[off[l][k] + r * 32 .. + 31] = the space-time plane (k, t) interpolated at the frame's timestamp, row r along axis k.
struct LineTab {
    unsigned off[4][3];                              // float offset of line (level, axis) inside the table
};

__constant__ int kCombA[6] = {0, 0, 0, 1, 1, 2};
__constant__ int kCombB[6] = {1, 2, 3, 2, 3, 3};

__device__ __forceinline__ void norm_coords(const HexArgs& a, const float* __restrict__ xyz, int g, float c[4])
{
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = (xyz[3 * g + k] - a.a0[k]) * (2.0f / (a.a1[k] - a.a0[k])) - 1.0f;
    c[3] = a.times ? a.times[g] : a.time;
}

// ---- helpers shared by the chunked kernels -------------------------------------------------------------------
__device__ __forceinline__ int time_sample(float c, int size, int& i0, int& i1, float& w0, float& w1)
{
    float gm;
    const float v = unnorm_clip(c, size, gm);
    const int x0 = (int)floorf(v), x1 = x0 + 1;
    w0 = (float)x1 - v;
    w1 = v - (float)x0;
    i0 = (x0 >= 0 && x0 < size) ? x0 : -1;
    i1 = (x1 >= 0 && x1 < size) ? x1 : -1;
    return 0;
}

}  // namespace

static void fill_args(const MomHexPlane* hp, int P, const float* times, float time, const uint32_t* order, bool grads, HexArgs* a)
{
    a->P = P; a->levels = hp->levels; a->time = time; a->times = times; a->order = order;
    for (int l = 0; l < 4; l++)
        for (int k = 0; k < 4; k++) a->res[l][k] = hp->res[l][k];
    for (int l = 0; l < 4; l++)
        for (int p = 0; p < 6; p++) { a->planes[l][p] = hp->planes[l][p]; a->grads[l][p] = grads ? hp->grads[l][p] : nullptr; }
    for (int k = 0; k < 3; k++) { a->a0[k] = hp->aabb[k]; a->a1[k] = hp->aabb[3 + k]; }
}

static int line_table(const MomHexPlane* hp, LineTab* lt)
{
    unsigned off = 0;
    for (int l = 0; l < 4; l++)
        for (int k = 0; k < 3; k++) {
            lt->off[l][k] = off;
            if (l < hp->levels) off += (unsigned)hp->res[l][k] * 32u;
        }
    return (int)off;
}
