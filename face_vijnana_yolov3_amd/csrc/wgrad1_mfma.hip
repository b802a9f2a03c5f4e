// Weight-gradient of the 1x1 layer with 64 input and 32 output channels (conv_2, reference yd.py:224) as a streaming kernel.
//
//   dw[n][c] += sum over pixels m of  dy[m][n] * x[m][c]            (n < 32, c < 64)
//
// HBM-bound: 664 MB of operands against 7 GFLOP.  The generic kernel (wgrad_mfma.hip, one 32x64 tile, all parallelism from the
// pixel split) stages 32 pixels per barrier -- 8 MFMAs per wave between two barriers: 0.24 ms (2.9 TB/s).  Here a 4-wave
// workgroup takes units of 128 consecutive pixels: the dy rows (128 x 32) and x rows (128 x 64) are prefetched into registers
// during the previous unit and staged once; a wave multiplies its 32 pixels (16 k-pairs x 2 MFMAs: the two 32-channel halves
// of x) and keeps its own 32 x 64 partial tile, added with float atomics when the workgroup has walked its unit range.
#include "conv.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int UP = 128;                      // pixels per unit, 32 per wave
constexpr int NTH = 256;
constexpr int CN = 32, CC = 64;
constexpr int LDY = CN + 4, LDX = CC + 4;

__global__ __launch_bounds__(NTH, 3) void wgrad1_kernel(const FvWgradArgs a, int n_units) {
    constexpr int NDY = UP * CN / 4 / NTH;       // 4 float4 per thread
    constexpr int NX = UP * CC / 4 / NTH;        // 8
    __shared__ __attribute__((aligned(16))) float dy_l[UP * LDY];
    __shared__ __attribute__((aligned(16))) float x_l[UP * LDX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int u_begin = (int)((long long)blockIdx.x * n_units / gridDim.x);
    const int u_end = (int)((long long)(blockIdx.x + 1) * n_units / gridDim.x);
    if (u_begin >= u_end) return;

    // rows >= M lie past num_records: the buffer descriptor returns zeros
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((unsigned)a.M * CC * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((unsigned)a.M * a.Ndy * 4u), 0x00020000);

    unsigned dy_rel[NDY], x_rel[NX];
#pragma unroll
    for (int p = 0; p < NDY; ++p) { const int f = tid + NTH * p; dy_rel[p] = (unsigned)((f >> 3) * a.Ndy + (f & 7) * 4) * 4u; }
#pragma unroll
    for (int p = 0; p < NX; ++p) { const int f = tid + NTH * p; x_rel[p] = (unsigned)((f >> 4) * CC + (f & 15) * 4) * 4u; }

    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

    u32x4 ry[NDY], rx[NX];
    auto issue = [&](int u) {
        const unsigned by = (unsigned)u * UP * (unsigned)a.Ndy * 4u, bx = (unsigned)u * UP * CC * 4u;
#pragma unroll
        for (int p = 0; p < NDY; ++p) ry[p] = __builtin_amdgcn_raw_buffer_load_b128(yr, by + dy_rel[p], 0, 0);
#pragma unroll
        for (int p = 0; p < NX; ++p) rx[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, bx + x_rel[p], 0, 0);
    };
    auto stage = [&]() {
#pragma unroll
        for (int p = 0; p < NDY; ++p) { const int f = tid + NTH * p; *reinterpret_cast<u32x4*>(&dy_l[(f >> 3) * LDY + (f & 7) * 4]) = ry[p]; }
#pragma unroll
        for (int p = 0; p < NX; ++p) { const int f = tid + NTH * p; *reinterpret_cast<u32x4*>(&x_l[(f >> 4) * LDX + (f & 15) * 4]) = rx[p]; }
    };
    const float* pa = dy_l + (wave * 32 + half) * LDY + l31;
    const float* pb = x_l + (wave * 32 + half) * LDX + l31;

    issue(u_begin);
    stage();
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
        const bool more = u + 1 < u_end;
        if (more) issue(u + 1);
        float fa0 = pa[0], fb0 = pb[0], fc0 = pb[32], fa1, fb1, fc1;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            fa1 = pa[2 * (j + 1) * LDY]; fb1 = pb[2 * (j + 1) * LDX]; fc1 = pb[2 * (j + 1) * LDX + 32];
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fc0, acc[1], 0, 0, 0);
            if (j + 2 < 16) { fa0 = pa[2 * (j + 2) * LDY]; fb0 = pb[2 * (j + 2) * LDX]; fc0 = pb[2 * (j + 2) * LDX + 32]; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fc1, acc[1], 0, 0, 0);
        }
        __syncthreads();
        if (more) stage();
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
            atomicAdd(a.dw + (size_t)n * CC + q * 32 + l31, acc[q][r]);
        }
}

}  // namespace

bool fv_wgrad1_ok(const FvWgradArgs& a) {
    return a.Cin == CC && a.N == CN && a.Ndy >= CN && (a.Ndy & 3) == 0 && a.Tw == 1 && a.taps.n == 1 && a.is == 1 && a.Hl == a.Hin &&
           a.Wl == a.Win && a.taps.dh[0] == 0 && a.taps.dw[0] == 0;
}

int fv_wgrad1_launch(fv_ctx* ctx, const FvWgradArgs& a) {
    const long long n_units = ((long long)a.M + UP - 1) / UP;
    FV_REQUIRE(ctx, n_units < (1ll << 24), "wgrad1: too many units");
    const int grid = n_units < 768 ? (int)n_units : 768;   // three workgroups per CU
    FvProfScope ps(ctx, "wgrad1_kernel", a.alg_flops, 4.0 * ((double)a.M * a.Cin + (double)a.M * a.N + (double)a.N * a.Cin));
    hipLaunchKernelGGL(wgrad1_kernel, dim3(grid), dim3(NTH), 0, ctx->stream, a, (int)n_units);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
