// Weight-gradient of the 3x3 layers with 32 input and 64 output channels (conv_1: stride 2, conv_3: stride 1 -- the two
// largest-spatial 3x3 layers of Darknet-53, reference yd.py:221-229) with the nine taps fused.
//
//   dw[n][r*3+q][c] += sum over output pixels (b, oh, ow) of  dy[b, oh, ow, n] * x[b, oh*S + r - 1, ow*S + q - 1, c]
//
// The generic kernel (wgrad_mfma.hip, 64x32 tiles) runs one workgroup per tap, so every tap stages its own shifted copy of
// the x rows and the whole dy chunk again, and a 64x32 tile pays the staging instructions of a chunk for 8 MFMAs per wave:
// 85 TF.  Here a workgroup takes a unit of 8 x 16 output pixels, stages its dy tile (128 pixels x 64 channels) and the x halo
// ((7 S + 3) x (15 S + 3) pixels x 32 channels) in LDS ONCE, and all nine taps read the halo at shifted addresses:
//   * 12 waves = (output-channel half) x (tap row r) x (upper / lower four rows of the unit); a wave keeps the three 32x32
//     accumulators of its row's taps q = 0..2.  (Twelve equal waves = three per SIMD; the first version, 6 waves per
//     workgroup and two workgroups per CU, left the SIMDs with 2+2+1+1 waves of a workgroup: 90 / 108 TF)
//   * K = pixels: a k-pair is two horizontally adjacent pixels (lanes 0-31 / 32-63), every LDS fragment address is a
//     per-lane constant plus a compile-time offset (the unit loop body is fully unrolled: 32 k-pairs x 3 MFMAs per wave)
//   * the next unit's rows are loaded into registers while the current one is multiplied, and stored to LDS between two
//     barriers; out-of-image halo pixels and pixels beyond the lattice come back as zeros from the buffer descriptor
//   * a workgroup walks a contiguous range of units (grid = 256 workgroups, one per CU) and adds its 64 x 288
//     partial tile with float atomics at the end, like the generic kernel (the caller zeroes dw once per step)
#include "conv.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int UR = 8, UC = 16;               // unit: 8 output rows x 16 output columns
constexpr int UP = UR * UC;                  // 128 pixels; a wave multiplies 64 of them = 32 k-pairs
constexpr int NTH = 768;                     // 12 waves
constexpr int CN = 64, CC = 32;              // dy channels used, x channels
constexpr int LDY = CN + 4, LDX = CC + 4;    // LDS row strides in floats (+4: 16-byte aligned rows, staggered banks)

template <int S>
__global__ __launch_bounds__(NTH, 1) void wgrad9_kernel(const FvWgradArgs a, int units_w, int units_h, int n_units) {
    constexpr int HR = (UR - 1) * S + 3, HC = (UC - 1) * S + 3;      // halo rows / columns: 10 x 18 (S = 1), 17 x 33 (S = 2)
    constexpr int NDY = (UP * CN / 4 + NTH - 1) / NTH;               // float4 loads per thread: dy tile (3)
    constexpr int NXF = HR * HC * CC / 4;                            // float4s of the halo
    constexpr int NX = (NXF + NTH - 1) / NTH;                        // (2 / 6)
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) float dy_l[UP * LDY];
    __shared__ __attribute__((aligned(16))) float x_l[HR * HC * LDX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nt = wave & 1, tr = (wave >> 1) % 3, ph = wave / 6;    // output-channel half, tap row, pixel half (rows 0-3 / 4-7)

    const int u_begin = (int)((long long)blockIdx.x * n_units / gridDim.x);
    const int u_end = (int)((long long)(blockIdx.x + 1) * n_units / gridDim.x);
    if (u_begin >= u_end) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.B * a.Hin * a.Win * CC * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.M * a.Ndy * 4u), 0x00020000);

    // per-thread constants of the staging slots
    // (row, column) packed as row << 8 | column; slots beyond the tile get row 2^12: never valid
    unsigned dy_rel[NDY]; int dy_rc[NDY];
#pragma unroll
    for (int p = 0; p < NDY; ++p) {
        const int f = tid + NTH * p, px = f >> 4, c4 = f & 15;
        dy_rc[p] = ((f < UP * CN / 4 ? px >> 4 : 1 << 12) << 8) | (px & 15);
        dy_rel[p] = (unsigned)(((px >> 4) * a.Wl + (px & 15)) * a.Ndy + c4 * 4) * 4u;
    }
    unsigned x_rel[NX]; int x_rc[NX];
#pragma unroll
    for (int p = 0; p < NX; ++p) {
        const int f = tid + NTH * p, hp = f >> 3, c4 = f & 7;
        const int hr = hp / HC, hc = hp - hr * HC;
        x_rc[p] = ((f < NXF ? hr : 1 << 12) << 8) | hc;
        x_rel[p] = (unsigned)((hr * a.Win + hc) * CC + c4 * 4) * 4u;
    }

    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

    u32x4 ry[NDY], rx[NX];
    // unit u -> image, first output row / column; loads of its dy tile and x halo into registers
    auto issue = [&](int u) {
        const int uc = u % units_w, t = u / units_w, ur = t % units_h, b = t / units_h;
        const int oh0 = ur * UR, ow0 = uc * UC;
        const int ih0 = oh0 * S - 1, iw0 = ow0 * S - 1;
        const unsigned base_y = (unsigned)(((b * a.Hl + oh0) * a.Wl + ow0) * a.Ndy) * 4u;
        const unsigned base_x = (unsigned)(((b * a.Hin + ih0) * a.Win + iw0) * CC) * 4u;      // modular when ih0 / iw0 = -1
        const int lim_r = a.Hl - oh0, lim_c = a.Wl - ow0;
#pragma unroll
        for (int p = 0; p < NDY; ++p) {
            const bool ok = ((dy_rc[p] >> 8) < lim_r) & ((dy_rc[p] & 255) < lim_c);
            ry[p] = __builtin_amdgcn_raw_buffer_load_b128(yr, ok ? base_y + dy_rel[p] : OOB, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const bool ok = ((unsigned)(ih0 + (x_rc[p] >> 8)) < (unsigned)a.Hin) & ((unsigned)(iw0 + (x_rc[p] & 255)) < (unsigned)a.Win);
            rx[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? base_x + x_rel[p] : OOB, 0, 0);
        }
    };
    auto stage = [&]() {     // (LDS addresses recomputed here: two fewer registers per slot than keeping them)
#pragma unroll
        for (int p = 0; p < NDY; ++p) {
            const int f = tid + NTH * p;
            if (NTH * p + NTH <= UP * CN / 4 || f < UP * CN / 4) *reinterpret_cast<u32x4*>(&dy_l[(f >> 4) * LDY + (f & 15) * 4]) = ry[p];
        }
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const int f = tid + NTH * p;
            if (NTH * p + NTH <= NXF || f < NXF) *reinterpret_cast<u32x4*>(&x_l[(f >> 3) * LDX + (f & 7) * 4]) = rx[p];
        }
    };

    // fragment addresses: A' = dy_l[pixel][nt*32 + l31], B'_q = x_l[halo pixel of (pixel, tap (tr, q))][l31]; pixel = 2 j + half
    const float* pa = dy_l + (ph * (UP / 2) + half) * LDY + nt * 32 + l31;
    const float* pb = x_l + ((ph * (UR / 2) * S + tr) * HC + half * S) * LDX + l31;
    auto frag = [&](int j, float& fa, float (&fb)[3]) {
        const int row = j >> 3, col = (2 * j) & 15;
        fa = pa[2 * j * LDY];
#pragma unroll
        for (int q = 0; q < 3; ++q) fb[q] = pb[((row * S) * HC + col * S + q) * LDX];
    };

    issue(u_begin);
    stage();
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
        const bool more = u + 1 < u_end;
        if (more) issue(u + 1);
        float fa0, fb0[3], fa1, fb1[3];
        frag(0, fa0, fb0);
#pragma unroll
        for (int j = 0; j < UP / 4; j += 2) {
            frag(j + 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0[q], acc[q], 0, 0, 0);
            if (j + 2 < UP / 4) frag(j + 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1[q], acc[q], 0, 0, 0);
        }
        __syncthreads();                 // every wave is done with this unit's tiles
        if (more) stage();
        __syncthreads();
    }

#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            atomicAdd(a.dw + ((size_t)n * 9 + tr * 3 + q) * CC + l31, acc[q][r]);
        }
}

}  // namespace

bool fv_wgrad9_ok(const FvWgradArgs& a) {
    if (a.Cin != CC || a.N != CN || a.Ndy < CN || (a.Ndy & 3) || a.Tw != 9 || a.taps.n != 9) return false;
    if (a.is != 1 && a.is != 2) return false;
    if (a.Hl * a.is != a.Hin || a.Wl * a.is != a.Win) return false;
    for (int t = 0; t < 9; ++t)
        if (a.taps.dh[t] != t / 3 - 1 || a.taps.dw[t] != t % 3 - 1 || a.taps.wslot[t] != t) return false;
    return true;
}

int fv_wgrad9_launch(fv_ctx* ctx, const FvWgradArgs& a) {
    const int units_w = (a.Wl + UC - 1) / UC, units_h = (a.Hl + UR - 1) / UR;
    const long long n_units = (long long)a.B * units_h * units_w;
    FV_REQUIRE(ctx, n_units < (1ll << 30), "wgrad9: too many units");
    const int grid = n_units < 256 ? (int)n_units : 256;   // one workgroup per CU, each with a contiguous range of units
    FvProfScope ps(ctx, a.is == 1 ? "wgrad9_kernel<1>" : "wgrad9_kernel<2>", a.alg_flops,
                   4.0 * ((double)a.B * a.Hin * a.Win * a.Cin + (double)a.M * a.N + (double)a.N * a.Tw * a.Cin));
    if (a.is == 1)
        hipLaunchKernelGGL(wgrad9_kernel<1>, dim3(grid), dim3(NTH), 0, ctx->stream, a, units_w, units_h, (int)n_units);
    else
        hipLaunchKernelGGL(wgrad9_kernel<2>, dim3(grid), dim3(NTH), 0, ctx->stream, a, units_w, units_h, (int)n_units);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
