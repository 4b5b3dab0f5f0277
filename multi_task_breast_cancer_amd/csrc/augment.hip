// Training-time augmentation of the joint (mask, image) stack on the GPU.
//
// Replaces the per-item CPU transforms of training_multitask.py:193-197 applied at src/dataset/BUSI_dataset.py:142-147:
//     RandomHorizontalFlip(p=0.5) -> RandomVerticalFlip(p=0.5) -> RandomRotation(degrees=360)
// (torchvision: nearest interpolation, expand=False, centre = image centre, fill 0), which at >1000 images/s per GPU
// is the loader's bottleneck.  One gather per output pixel: torchvision rotates by building an affine grid
// (x_o = col - W/2 + 1/2, theta = [[cos a, -sin a, 0], [sin a, cos a, 0]] scaled by 1/(W/2), 1/(H/2)) and sampling it
// with grid_sample(mode=nearest, padding zeros, align_corners=False): pixel = nearbyint(((g + 1) * size - 1) / 2).
// The same arithmetic, in the same order, is evaluated here; the flips are folded into the source index.  The
// per-sample parameters (cos, sin, flip_h, flip_v) come from the host (multi_task_breast_cancer_amd/augment.py).
// HBM-bound: reads <= C*H*W*4 bytes, writes C*H*W*4 bytes per sample.
#include "common.h"

namespace {

struct AugP {
    int N, C, H, W;
    const float* src;
    float* dst;
    const float* params;      // (N, 4): cos a, sin a, flip_h (0/1), flip_v (0/1)
};

__global__ void augment_flip_rotate_kernel(const AugP p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long HW = (long long)p.H * p.W;
    if (idx >= (long long)p.N * HW) return;
    const int n = (int)(idx / HW);
    const int rem = (int)(idx % HW), y = rem / p.W, x = rem % p.W;
    const float ca = p.params[4 * n], sa = p.params[4 * n + 1];
    const bool fh = p.params[4 * n + 2] != 0.f, fv = p.params[4 * n + 3] != 0.f;
    const float xo = (float)x - 0.5f * (float)p.W + 0.5f, yo = (float)y - 0.5f * (float)p.H + 0.5f;
    const float hw = 0.5f * (float)p.W, hh = 0.5f * (float)p.H;
    const float gx = fmaf(yo, -sa / hw, xo * (ca / hw));
    const float gy = fmaf(yo, ca / hh, xo * (sa / hh));
    const float ix = ((gx + 1.f) * (float)p.W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)p.H - 1.f) * 0.5f;
    const float rx = nearbyintf(ix), ry = nearbyintf(iy);
    const bool inb = rx >= 0.f && rx < (float)p.W && ry >= 0.f && ry < (float)p.H;
    int xs = inb ? (int)rx : 0, ys = inb ? (int)ry : 0;
    if (fh) xs = p.W - 1 - xs;
    if (fv) ys = p.H - 1 - ys;
    const float* s = p.src + (size_t)n * p.C * HW + (size_t)ys * p.W + xs;
    float* d = p.dst + (size_t)n * p.C * HW + rem;
    for (int c = 0; c < p.C; ++c) d[(size_t)c * HW] = inb ? s[(size_t)c * HW] : 0.f;
}

}  // namespace

extern "C" int mtbc_augment_flip_rotate(const float* src, float* dst, const float* params, int32_t N, int32_t C, int32_t H,
                                        int32_t W, void* stream) {
    if (!src || !dst || !params) return MTBC_E_BADARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return MTBC_E_BADSHAPE;
    if (src == dst) return MTBC_E_BADARG;             // a gather cannot run in place
    AugP p{N, C, H, W, src, dst, params};
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(augment_flip_rotate_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
