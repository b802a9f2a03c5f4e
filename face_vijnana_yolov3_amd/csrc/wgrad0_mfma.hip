// Weight-gradient of the first layer (3 -> 32 channels, 3x3, stride 1, pad 1; reference yd.py:221), the weight-gradient
// member of the halo-tile family (conv0_direct.hip is its forward).
//
//   dw[n][(r*3+q)*3 + c] += sum over pixels (b, h, w) of  dy[b, h, w, n] * x[b, h + r - 1, w + q - 1, c]        (27 of 32 K slots)
//
// HBM-bound: 886 MB of dy against 12 GFLOP.  The generic gather kernel (wgrad_mfma.hip) stages 32 pixels per barrier and
// gathers the 27 patch values of a pixel with four scalar loads per lane: 0.44 ms (2.2 TB/s).  Here a 4-wave workgroup takes
// units of 8 x 32 pixels: the dy tile (256 pixels x 32 channels) and the x halo (10 x 34 pixels x 3 channels, a zero pad word
// per pixel) are prefetched into registers during the previous unit and staged once; a wave multiplies its 64 pixels --
// 32 k-pairs, one MFMA each: A' = dy[pixel][n], B' = the lane's patch slot (tap, channel) read from the halo at a per-lane
// constant + compile-time offset (slots 27..31 read the zero pad word).  Waves never exchange partial sums: each adds its
// own 32 x 27 tile with float atomics when its workgroup has walked its unit range (the caller zeroes dw once per step).
#include "conv.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int UR = 8, UC = 32;               // unit: 8 rows x 32 columns = 256 pixels, 64 per wave
constexpr int NTH = 256;
constexpr int CN = 32, CI = 3;               // dy channels, x channels
constexpr int LDY = CN + 4;                  // dy_l row stride (floats)
constexpr int HR = UR + 2, HC = UC + 2;      // halo 10 x 34 pixels, 4 floats each (3 channels + a zero word)

__global__ __launch_bounds__(NTH, 3) void wgrad0_kernel(const FvWgradArgs a, int units_w, int units_h, int n_units) {
    constexpr int NDY = UR * UC * CN / 4 / NTH;                      // float4 loads per thread: dy tile (8)
    constexpr int NXW = HR * HC * CI;                                // words of the halo (1020)
    constexpr int NX = (NXW + NTH - 1) / NTH;                        // (4)
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) float dy_l[UR * UC * LDY];
    __shared__ __attribute__((aligned(16))) float x_l[HR * HC * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    const int u_begin = (int)((long long)blockIdx.x * n_units / gridDim.x);
    const int u_end = (int)((long long)(blockIdx.x + 1) * n_units / gridDim.x);
    if (u_begin >= u_end) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.B * a.Hin * a.Win * CI * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.M * a.Ndy * 4u), 0x00020000);

    // zero pad word of every halo pixel (never overwritten)
    for (int i = tid; i < HR * HC; i += NTH) x_l[i * 4 + 3] = 0.0f;

    unsigned dy_rel[NDY]; int dy_rc[NDY];        // slot p: pixel (row << 8 | col) of the unit, byte offset relative to the unit origin
#pragma unroll
    for (int p = 0; p < NDY; ++p) {
        const int f = tid + NTH * p, px = f >> 3, c4 = f & 7;
        dy_rc[p] = ((px >> 5) << 8) | (px & 31);
        dy_rel[p] = (unsigned)(((px >> 5) * a.Wl + (px & 31)) * a.Ndy + c4 * 4) * 4u;
    }
    unsigned x_rel[NX]; int x_rc[NX], x_lds[NX];  // halo word w = (row*HC + col)*3 + c
#pragma unroll
    for (int p = 0; p < NX; ++p) {
        const int w = tid + NTH * p, hp = w / CI, c = w - hp * CI;
        const int hr = hp / HC, hc = hp - hr * HC;
        x_rc[p] = ((w < NXW ? hr : 1 << 12) << 8) | hc;
        x_rel[p] = (unsigned)((hr * a.Win + hc) * CI + c) * 4u;
        x_lds[p] = (w < NXW ? hp : 0) * 4 + c;
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    u32x4 ry[NDY];
    unsigned rx[NX];
    auto issue = [&](int u) {
        const int uc = u % units_w, t = u / units_w, ur = t % units_h, b = t / units_h;
        const int h0 = ur * UR, w0 = uc * UC;
        const unsigned base_y = (unsigned)(((b * a.Hl + h0) * a.Wl + w0) * a.Ndy) * 4u;
        const unsigned base_x = (unsigned)(((b * a.Hin + h0 - 1) * a.Win + w0 - 1) * CI) * 4u;      // modular at the image border
        const int lim_r = a.Hl - h0, lim_c = a.Wl - w0;
#pragma unroll
        for (int p = 0; p < NDY; ++p) {
            const bool ok = ((dy_rc[p] >> 8) < lim_r) & ((dy_rc[p] & 255) < lim_c);
            ry[p] = __builtin_amdgcn_raw_buffer_load_b128(yr, ok ? base_y + dy_rel[p] : OOB, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const bool ok = ((unsigned)(h0 - 1 + (x_rc[p] >> 8)) < (unsigned)a.Hin) & ((unsigned)(w0 - 1 + (x_rc[p] & 255)) < (unsigned)a.Win);
            rx[p] = __builtin_amdgcn_raw_buffer_load_b32(xr, ok ? base_x + x_rel[p] : OOB, 0, 0);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int p = 0; p < NDY; ++p) {
            const int f = tid + NTH * p;
            *reinterpret_cast<u32x4*>(&dy_l[(f >> 3) * LDY + (f & 7) * 4]) = ry[p];
        }
#pragma unroll
        for (int p = 0; p < NX; ++p)
            if (NTH * p + NTH <= NXW || tid + NTH * p < NXW) x_l[x_lds[p]] = __uint_as_float(rx[p]);
    };

    // fragments: pixel = wave*64 + 2 j + half = (row wave*2 + (j >> 4), column (2 j & 31) + half)
    const int ktap = l31 / CI, kc = l31 - ktap * CI;                          // lane's patch slot: tap, channel (slots >= 27: zero word)
    const int kofs = l31 < 27 ? ((ktap / 3) * HC + ktap % 3) * 4 + kc : 3;
    const float* pa = dy_l + (wave * 64 + half) * LDY + l31;
    const float* pb = x_l + ((wave * 2) * HC + half) * 4 + kofs;

    issue(u_begin);
    stage();
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
        const bool more = u + 1 < u_end;
        if (more) issue(u + 1);
        float fa0 = pa[0], fb0 = pb[0], fa1, fb1;
#pragma unroll
        for (int j = 0; j < 32; j += 2) {
            fa1 = pa[2 * (j + 1) * LDY]; fb1 = pb[(((j + 1) >> 4) * HC + ((2 * (j + 1)) & 31)) * 4];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc, 0, 0, 0);
            if (j + 2 < 32) { fa0 = pa[2 * (j + 2) * LDY]; fb0 = pb[(((j + 2) >> 4) * HC + ((2 * (j + 2)) & 31)) * 4]; }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc, 0, 0, 0);
        }
        __syncthreads();
        if (more) stage();
        __syncthreads();
    }

    if (l31 < 27) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
            atomicAdd(a.dw + (size_t)n * 27 + l31, acc[r]);
        }
    }
}

}  // namespace

bool fv_wgrad0_ok(const FvWgradArgs& a) {
    return a.Cin == CI && a.N == CN && a.Ndy >= CN && (a.Ndy & 3) == 0 && a.is == 1 && a.Hl == a.Hin && a.Wl == a.Win && a.taps.n == 9;
}

int fv_wgrad0_launch(fv_ctx* ctx, const FvWgradArgs& a) {
    const int units_w = (a.Wl + UC - 1) / UC, units_h = (a.Hl + UR - 1) / UR;
    const long long n_units = (long long)a.B * units_h * units_w;
    FV_REQUIRE(ctx, n_units < (1ll << 30), "wgrad0: too many units");
    const int grid = n_units < 768 ? (int)n_units : 768;   // three workgroups per CU, contiguous unit ranges
    FvProfScope ps(ctx, "wgrad0_kernel", a.alg_flops,
                   4.0 * ((double)a.B * a.Hin * a.Win * a.Cin + (double)a.M * a.N + (double)a.N * 27));
    hipLaunchKernelGGL(wgrad0_kernel, dim3(grid), dim3(NTH), 0, ctx->stream, a, units_w, units_h, (int)n_units);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
