// Training-mode forward of the 3x3 layers with 32 input and 64 output channels (conv_1: stride 2, conv_3: stride 1; reference
// yd.py:221-229) from an LDS halo tile -- the forward counterpart of wgrad9_mfma.hip.
//
//   z[b, oh, ow, n] = sum over taps (r, q) and channels c of  x[b, oh*S + r - 1, ow*S + q - 1, c] * w[n][r*3+q][c]
//   + per-channel sum / sum of squares of z added to the fp64 statistics slots (FV_EPI_STATS with stat_slots)
//
// In the generic kernel (conv_mfma.hip, 128x64 tiles) these two launches have 13 520 tiles of only nine K steps each: the
// per-tile prologue / epilogue and the staging of a shifted 16 KB A tile per tap leave them at 95-110 TF.  Here one workgroup
// per CU keeps ALL weights (64 x 288 floats) in LDS for its whole life and walks units of 8 x 16 output pixels: the x halo
// ((7 S + 3) x (15 S + 3) pixels x 32 channels) is staged once per unit, the nine taps read it at shifted compile-time
// offsets, the next unit's halo is prefetched into registers during the multiplication.  8 waves = 4 pixel row-pairs x 2
// channel halves, one 32x32 accumulator each; K order = the generic kernel's (tap-major, inside every 8 channels
// 0,4,1,5,2,6,3,7), so z is BIT-IDENTICAL to conv_kernel<64,...>.  The statistics are summed per unit in fp32, across units
// in fp64 registers, and added to one slot per workgroup at the end (the generic kernel adds per tile).
#include "conv.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int UR = 8, UC = 16;               // unit: 8 output rows x 16 output columns = 128 pixels = 4 row-pairs of 32
constexpr int NTH = 512;                     // 8 waves
constexpr int CC = 32, CN = 64, KW = 9 * CC; // x channels, output channels, K
constexpr int LDX = CC + 4, LDW = KW + 4;    // LDS row strides (floats): 36 mod 64 keeps the ds_read_b128 groups conflict-free

template <int S>
__global__ __launch_bounds__(NTH, 1) void conv9_fwd_kernel(const FvConvArgs a, int units_w, int units_h, int n_units) {
    constexpr int HR = (UR - 1) * S + 3, HC = (UC - 1) * S + 3;      // halo: 10 x 18 (S = 1), 17 x 33 (S = 2)
    constexpr int NXF = HR * HC * CC / 4;                            // float4s of the halo
    constexpr int NX = (NXF + NTH - 1) / NTH;                        // per-thread staging slots (3 / 9)
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) float w_l[CN * LDW];
    __shared__ __attribute__((aligned(16))) float x_l[HR * HC * LDX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nt = wave & 1, rt = wave >> 1;                         // channel half, pixel row-pair

    const int u_begin = (int)((long long)blockIdx.x * n_units / gridDim.x);
    const int u_end = (int)((long long)(blockIdx.x + 1) * n_units / gridDim.x);
    if (u_begin >= u_end) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.B * a.Hin * a.Win * CC * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.out, 0, (int)((unsigned)a.M * CN * 4u), 0x00020000);

    // all weights -> LDS, [n][tap*32 + c] with padded rows
#pragma unroll
    for (int p = 0; p < CN * KW / 4 / NTH; ++p) {
        const int f = tid + NTH * p, n = f / (KW / 4), k4 = f - n * (KW / 4);
        *reinterpret_cast<float4*>(&w_l[n * LDW + k4 * 4]) = *reinterpret_cast<const float4*>(a.w + (size_t)n * KW + k4 * 4);
    }

    unsigned x_rel[NX]; int x_rc[NX];       // halo slot: byte offset relative to the halo origin, (row << 8 | column); row 2^12 = no slot
#pragma unroll
    for (int p = 0; p < NX; ++p) {
        const int f = tid + NTH * p, hp = f >> 3, c4 = f & 7;
        const int hr = hp / HC, hc = hp - hr * HC;
        x_rc[p] = ((f < NXF ? hr : 1 << 12) << 8) | hc;
        x_rel[p] = (unsigned)((hr * a.Win + hc) * CC + c4 * 4) * 4u;
    }
    u32x4 rx[NX];
    auto issue = [&](int u) {
        const int uc = u % units_w, t = u / units_w, ur = t % units_h, b = t / units_h;
        const int ih0 = ur * UR * S - 1, iw0 = uc * UC * S - 1;
        const unsigned base_x = (unsigned)(((b * a.Hin + ih0) * a.Win + iw0) * CC) * 4u;      // modular when ih0 / iw0 = -1
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const bool ok = ((unsigned)(ih0 + (x_rc[p] >> 8)) < (unsigned)a.Hin) & ((unsigned)(iw0 + (x_rc[p] & 255)) < (unsigned)a.Win);
            rx[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? base_x + x_rel[p] : OOB, 0, 0);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const int f = tid + NTH * p;
            if (NTH * p + NTH <= NXF || f < NXF) *reinterpret_cast<u32x4*>(&x_l[(f >> 3) * LDX + (f & 7) * 4]) = rx[p];
        }
    };

    // fragments: A = x_l[halo pixel of (output pixel, tap)][8 g + 4 half ..], B = w_l[n][tap*32 + 8 g + 4 half ..]
    const float* pa = x_l + (((rt * 2 + (l31 >> 4)) * S) * HC + (l31 & 15) * S) * LDX + half * 4;
    const float* pb = w_l + (nt * 32 + l31) * LDW + half * 4;
    auto frag = [&](int i, float4& fa, float4& fb) {          // i = tap * 4 + g
        const int t = i >> 2, g = i & 3, tr = t / 3, tq = t - tr * 3;
        fa = *reinterpret_cast<const float4*>(pa + (tr * HC + tq) * LDX + g * 8);
        fb = *reinterpret_cast<const float4*>(pb + t * CC + g * 8);
    };
    // output: lane = channel nt*32 + l31; accumulator register r = pixel (r & 3) + 8 (r >> 2) + 4 half of the wave's 32
    //         = unit row rt*2 + (r >> 3), unit column 8 ((r >> 2) & 1) + (r & 3) + 4 half
    const int ocol = 4 * half, orow = rt * 2;
    const unsigned out_lane = (unsigned)((orow * a.Wl + ocol) * CN + nt * 32 + l31) * 4u;
    double sum_d = 0.0, sq_d = 0.0;

    issue(u_begin);
    stage();
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
        const bool more = u + 1 < u_end;
        if (more) issue(u + 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        float4 fa0, fb0, fa1, fb1;
        frag(0, fa0, fb0);
#pragma unroll
        for (int i = 0; i < 36; i += 2) {
            frag(i + 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.x, fb0.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.y, fb0.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.z, fb0.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.w, fb0.w, acc, 0, 0, 0);
            if (i + 2 < 36) frag(i + 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.x, fb1.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.y, fb1.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.z, fb1.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.w, fb1.w, acc, 0, 0, 0);
        }
        // store z, accumulate the statistics of the pixels that exist
        {
            const int uc = u % units_w, t = u / units_w, ur = t % units_h, b = t / units_h;
            const int oh0 = ur * UR, ow0 = uc * UC;
            const unsigned base_o = (unsigned)(((b * a.Hl + oh0) * a.Wl + ow0) * CN) * 4u + out_lane;
            const int lim_r = a.Hl - oh0 - orow, lim_c = a.Wl - ow0 - ocol;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r >> 3, col = 8 * ((r >> 2) & 1) + (r & 3);
                const bool ok = (row < lim_r) & (col < lim_c);
                const float v = ok ? acc[r] : 0.0f;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), orr, ok ? base_o + (unsigned)((row * a.Wl + col) * CN) * 4u : OOB, 0, 0);
                s += v; q += v * v;
            }
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            sum_d += (double)s; sq_d += (double)q;
        }
        __syncthreads();                 // every wave is done reading this unit's halo
        if (more) stage();
        __syncthreads();
    }

    // channel sums of the four row-pair waves -> one fp64 atomic per channel and workgroup
    double* red = reinterpret_cast<double*>(x_l);          // [2][8 waves][32] (free: the loop ended with a barrier)
    if (half == 0) { red[wave * 32 + l31] = sum_d; red[(8 + wave) * 32 + l31] = sq_d; }
    __syncthreads();
    if (tid < CN) {
        const int c = tid & 31, h = tid >> 5;              // waves with nt == h: h, h + 2, h + 4, h + 6
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s += red[(h + 2 * k) * 32 + c]; q += red[(8 + h + 2 * k) * 32 + c]; }
        double* sl = a.stat_slots + (size_t)(blockIdx.x % a.stat_nslot) * 2 * CN;
        unsafeAtomicAdd(sl + tid, s);
        unsafeAtomicAdd(sl + CN + tid, q);
    }
}

}  // namespace

bool fv_conv9_fwd_ok(const FvConvArgs& a) {
    if (a.Cin != CC || a.Nout != CN || a.Tw != 9 || a.nclass != 1 || a.ksplit > 1) return false;
    if (a.epi != FV_EPI_STATS || !a.stat_slots || a.stat_nslot < 1) return false;
    if ((a.is != 1 && a.is != 2) || a.os != 1 || a.Hout != a.Hl || a.Wout != a.Wl || a.oph[0] || a.opw[0]) return false;
    if (a.Hl * a.is != a.Hin || a.Wl * a.is != a.Win || a.taps[0].n != 9) return false;
    // masked edge stores use the 32-bit byte offset 0x80000000 as "outside": it must lie beyond the output tensor
    if ((long long)a.B * a.Hout * a.Wout * a.Nout * 4 >= (1ll << 31)) return false;
    for (int t = 0; t < 9; ++t)
        if (a.taps[0].dh[t] != t / 3 - 1 || a.taps[0].dw[t] != t % 3 - 1 || a.taps[0].wslot[t] != t) return false;
    return true;
}

int fv_conv9_fwd_launch(fv_ctx* ctx, const FvConvArgs& a) {
    const int units_w = (a.Wl + UC - 1) / UC, units_h = (a.Hl + UR - 1) / UR;
    const long long n_units = (long long)a.B * units_h * units_w;
    FV_REQUIRE(ctx, n_units < (1ll << 30), "conv9: too many units");
    const int grid = n_units < 256 ? (int)n_units : 256;   // one workgroup per CU (155 KB of LDS), contiguous unit ranges
    FvProfScope ps(ctx, a.is == 1 ? "conv9_fwd_kernel<1>" : "conv9_fwd_kernel<2>", a.alg_flops,
                   4.0 * ((double)a.B * a.Hin * a.Win * a.Cin + (double)a.Nout * a.Tw * a.Cin + (double)a.M * a.Nout));
    if (a.is == 1)
        hipLaunchKernelGGL(conv9_fwd_kernel<1>, dim3(grid), dim3(NTH), 0, ctx->stream, a, units_w, units_h, (int)n_units);
    else
        hipLaunchKernelGGL(conv9_fwd_kernel<2>, dim3(grid), dim3(NTH), 0, ctx->stream, a, units_w, units_h, (int)n_units);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
