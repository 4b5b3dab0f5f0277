// Experiment: HBM write rate of the igemm epilogue's store pattern (one instruction = 16 channel planes x 64 bytes)
// against a plane-major pattern (one instruction = 1 plane x 8 tile rows x 128 bytes) on the same tile walk.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/write_pattern_probe.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
// tile = 8 rows x 32 cols of a 256x256 plane, C channels, N images; block = 256 threads handles one tile x 32 channels
template <int MODE>
__global__ __launch_bounds__(256) void wr(float* out, int N, int C, int H, int W, int ntiles, int cblocks) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kg = lane >> 4;
    const int HW = H * W, tiles_x = W / 32, tiles_y = H / 8;
    const int per = (ntiles + 7) >> 3, xcd = blockIdx.x & 7, tstep = gridDim.x >> 3;
    const int tend = min(ntiles, (xcd + 1) * per);
    const int cb = blockIdx.y;
    for (int tile = xcd * per + (blockIdx.x >> 3); tile < tend; tile += tstep) {
        int t = tile; const int tx = t % tiles_x; t /= tiles_x; const int ty = t % tiles_y; t /= tiles_y;
        const int n = t, x0 = tx * 32, y0 = ty * 8;
        const f32x4 v = {1.f, 2.f, 3.f, (float)tile};
        if (MODE == 0) {            // epilogue pattern: lane (j, kg): channel 16m + j, 4 pixels at x = 16(g&1) + 4kg, row 2wv + (g>>1)
            for (int m = 0; m < 2; ++m) {
                float* cbp = out + ((size_t)n * C + cb * 32 + 16 * m + j) * HW;
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(cbp + (y0 + 2 * wv + (g >> 1)) * W + x0 + 16 * (g & 1) + 4 * kg) = v;
            }
        } else {                    // plane-major: wave wv owns channels 8wv..8wv+7 of the 32; one instruction = 1 plane x 8 rows x 128 B
            for (int c = 0; c < 8; ++c) {
                float* cbp = out + ((size_t)n * C + cb * 32 + 8 * wv + c) * HW;
                *reinterpret_cast<f32x4*>(cbp + (y0 + (lane >> 3)) * W + x0 + 4 * (lane & 7)) = v;
            }
        }
    }
}
int main() {
    const int N = 32, C = 96, H = 256, W = 256;        // 24..144-channel outputs of the step; 96 = 3 channel blocks
    float* d; CK(hipMalloc(&d, (size_t)N * C * H * W * 4));
    const int ntiles = N * (H / 8) * (W / 32), cblocks = C / 32;
    dim3 grid(256 * 3 / cblocks / 8 * 8, cblocks);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) { if (mode == 0) wr<0><<<grid, 256>>>(d, N, C, H, W, ntiles, cblocks); else wr<1><<<grid, 256>>>(d, N, C, H, W, ntiles, cblocks); }
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 10; ++rep) { if (mode == 0) wr<0><<<grid, 256>>>(d, N, C, H, W, ntiles, cblocks); else wr<1><<<grid, 256>>>(d, N, C, H, W, ntiles, cblocks); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("mode %d (%s): %.3f ms  %.2f TB/s\n", mode, mode ? "1 plane x 8 rows x 128 B per instruction" : "16 planes x 64 B per instruction", ms,
               (double)N * C * H * W * 4 / ms / 1e9);
    }
    return 0;
}
