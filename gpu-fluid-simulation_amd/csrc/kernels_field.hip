// kernels_field.hip — obstacle push-out field: HIP version of the reference's
// CPU `generate_smooth_gradient_field` (src/main.rs:403-515), which the renderer runs on a spawned
// thread per frame (src/renderer.rs:538-547) and uploads into force_field_texture (:497-502).
//
// The reference is a SEQUENTIAL two-pass raster propagation of "nearest source pixel" (forward:
// left, top-left, top, top-right; backward: the mirror), each pixel reading neighbours already
// updated in the same pass, first strict improvement wins.  Its result is not the exact Euclidean
// transform, so it is reproduced step for step: pixel (x, y) only depends on pixels with a smaller
// x + 2y, hence all pixels of one wavefront t = x + 2y are independent.  One thread per image row
// walks its row left to right (x = t - 2y); what it needs from the row above (x-1, x, x+1) was
// produced 3, 2 and 1 wavefronts earlier and sits in a 4-entry LDS ring; its own left neighbour is
// in a register.  One barrier per wavefront, no global loads inside the sweep.
#include "fs_device.h"
#include "fs_kernels.h"

namespace fsd {

#define FIELD_MAX_ROWS 1024

__device__ __forceinline__ float sqd(uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2) {   // main.rs:441-445
    const float dx = (float)x1 - (float)x2, dy = (float)y1 - (float)y2;
    return dx * dx + dy * dy;
}

// REVERSE = false: forward pass (main.rs:447-468).  REVERSE = true: backward pass (:470-491) run on
// mirrored coordinates (x -> w-1-x, y -> h-1-y), which turns it into the same sweep.
template <bool REVERSE>
__device__ __forceinline__ void field_sweep(uint32_t w, uint32_t h, float* __restrict__ dist, uint32_t* __restrict__ nearest,
                                            uint32_t (*s_ring)[4]) {
    const uint32_t ly = threadIdx.x;                       // logical row of this thread
    const bool row_ok = ly < h;
    const uint32_t py = REVERSE ? h - 1u - ly : ly;        // physical row
    uint32_t left_near = 0;                                // nearest[] of my (logical) left neighbour after its update
    const uint32_t waves = w + 2u * (h - 1u);
    for (uint32_t t = 0; t < waves; ++t) {
        const int32_t lxs = (int32_t)t - 2 * (int32_t)ly;
        const bool act = row_ok && lxs >= 0 && lxs < (int32_t)w;
        uint32_t mine = 0;
        if (act) {
            const uint32_t lx = (uint32_t)lxs;
            const uint32_t px = REVERSE ? w - 1u - lx : lx;
            const size_t idx = (size_t)py * w + px;
            float d = dist[idx];
            uint32_t nr = nearest[idx];
            // candidates in the reference's order; `cand` holds PHYSICAL source coordinates
            #pragma unroll
            for (int c = 0; c < 4; ++c) {
                // logical neighbour offsets: (-1,0), (-1,-1), (0,-1), (+1,-1)
                const int32_t nlx = (int32_t)lx + (c == 3 ? 1 : (c == 2 ? 0 : -1));
                const int32_t nly = (int32_t)ly - (c == 0 ? 0 : 1);
                if (nlx < 0 || nlx >= (int32_t)w || nly < 0) continue;      // `if nx < width && ny < height`
                const uint32_t cand = (c == 0) ? left_near : s_ring[ly - 1u][(uint32_t)nlx & 3u];
                const float cd = sqd(px, py, cand & 0xFFFFu, cand >> 16);
                if (cd < d) { d = cd; nr = cand; }
            }
            dist[idx] = d;
            nearest[idx] = nr;
            mine = nr;
            left_near = nr;
        }
        __syncthreads();                 // everyone has read the ring slots of this wavefront
        if (act) s_ring[ly][(uint32_t)lxs & 3u] = mine;
        __syncthreads();                 // ... and sees the new ones before the next wavefront
    }
}

// One workgroup (h <= 1024 threads).  image: u8 mask, > 128 = source (main.rs:416); without any
// source the image border is the source set (:426-438).
__global__ __launch_bounds__(FIELD_MAX_ROWS) void k_gradient_field(const unsigned char* __restrict__ image, uint32_t w, uint32_t h,
                                                                   float* __restrict__ dist, uint32_t* __restrict__ nearest,
                                                                   float2* __restrict__ field) {
    __shared__ uint32_t s_ring[FIELD_MAX_ROWS][4];
    __shared__ int s_has_white;
    const uint32_t y = threadIdx.x;
    if (y == 0) s_has_white = 0;
    __syncthreads();
    int white = 0;
    if (y < h)
        for (uint32_t x = 0; x < w; ++x) {
            const bool src = image[(size_t)y * w + x] > 128;
            dist[(size_t)y * w + x] = src ? 0.0f : 3.40282347e+38f;       // f32::MAX
            nearest[(size_t)y * w + x] = src ? (x | (y << 16)) : 0u;      // (0, 0) until reached
            white |= src ? 1 : 0;
        }
    if (white) atomicOr(&s_has_white, 1);
    __syncthreads();
    if (!s_has_white && y < h)
        for (uint32_t x = 0; x < w; ++x)
            if (y == h - 1u || y == 0u || x == w - 1u || x == 0u) {
                dist[(size_t)y * w + x] = 0.0f;
                nearest[(size_t)y * w + x] = x | (y << 16);
            }
    __syncthreads();
    field_sweep<false>(w, h, dist, nearest, s_ring);
    __syncthreads();
    field_sweep<true>(w, h, dist, nearest, s_ring);
    __syncthreads();
    if (y < h)
        for (uint32_t x = 0; x < w; ++x) {                                  // main.rs:496-511
            const uint32_t nr = nearest[(size_t)y * w + x];
            const float dx = (float)x - (float)(nr & 0xFFFFu), dy = (float)y - (float)(nr >> 16);
            const float len = sqrt_rn(dx * dx + dy * dy);
            const float gx = len > 1e-6f ? dx : 0.0f, gy = len > 1e-6f ? dy : 0.0f;
            field[(size_t)y * w + x] = make_float2(-gx, -gy);
        }
}

void launch_gradient_field(hipStream_t st, const unsigned char* image, uint32_t w, uint32_t h, float* dist,
                           uint32_t* nearest, float2* field) {
    hipLaunchKernelGGL(k_gradient_field, dim3(1), dim3(FIELD_MAX_ROWS), 0, st, image, w, h, dist, nearest, field);
}

}  // namespace fsd
