// Data-gradient of the stride-2 3x3 layer with 32 input and 64 output channels (conv_1, reference yd.py:222-223) from an LDS
// halo tile of dy, with the fused BatchNorm-backward reduction of the layer below (FV_EPI_BNRED) -- the third member of the
// conv9 / wgrad9 family.
//
//   dx[b, 2a+ph, 2c+pw, i] = sum over taps (r, q) with (ph+1-r), (pw+1-q) even, and channels n of
//                            dy[b, a + (ph+1-r)/2, c + (pw+1-q)/2, n] * w_t[i][r*3+q][n]
//
// The tile kernel runs the four parity classes (1 / 2 / 2 / 4 taps) as 54 080 tiles of two to eight K steps: prologue and
// epilogue dominate (73 TF).  Here one 8-wave workgroup per CU keeps the whole transposed weight image (32 x 576 floats) in
// LDS and walks units of 8 x 16 dy pixels (= 16 x 32 dx pixels): the dy halo (9 x 17 pixels x 64 channels) is staged once per
// unit, prefetched into registers during the previous unit.  Waves 0-3 (one per pixel row-pair) compute the 4-tap class,
// waves 4-7 the 1-, 2- and 2-tap classes one after the other: every SIMD gets one wave of each kind = nine tap-units.  K order
// inside a class = the tile kernel's (its tap list, then 0,4,1,5,2,6,3,7 inside every eight channels): dx is BIT-IDENTICAL
// to conv_kernel<32,4,1>.  z of the layer below is loaded while a class is multiplied; the per-channel sums of gy and
// gy * xhat are kept per lane (a lane owns one channel), in fp32 per class tile and in fp64 across tiles and units.
#include <type_traits>
#include "conv.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int UR = 8, UC = 16;               // unit: 8 x 16 dy pixels; a wave's 32x32 tile = 2 lattice rows x 16 columns
constexpr int NTH = 512;
constexpr int CD = 64, CX = 32, KW = 9 * CD; // dy channels, dx channels, K of the weight image
constexpr int LDD = CD + 4, LDW = KW + 4;    // LDS row strides in floats (68 and 580 = 36 mod 64 / 4 mod 64: b128 groups conflict-free)
constexpr int HR = UR + 1, HC = UC + 1;      // halo: one extra row below, one extra column to the right

__global__ __launch_bounds__(NTH, 1) void dgrad9s2_kernel(const FvConvArgs a, int units_w, int units_h, int n_units) {
    constexpr int NXF = HR * HC * CD / 4;                            // float4s of the halo (2448)
    constexpr int NX = (NXF + NTH - 1) / NTH;                        // staging slots per thread (5)
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) float w_l[CX * LDW];
    __shared__ __attribute__((aligned(16))) float d_l[HR * HC * LDD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int grp = wave >> 2, rt = wave & 3;                        // class group, lattice row-pair

    const int u_begin = (int)((long long)blockIdx.x * n_units / gridDim.x);
    const int u_end = (int)((long long)(blockIdx.x + 1) * n_units / gridDim.x);
    if (u_begin >= u_end) return;

    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.B * a.Hin * a.Win * a.Cin * 4u), 0x00020000);
    const unsigned out_bytes = (unsigned)a.B * a.Hout * a.Wout * CX * 4u;
    const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)out_bytes, 0x00020000);
    const bool bnred = (a.epi & FV_EPI_BNRED) != 0;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void*)(bnred ? a.bn_z : a.out), 0, (int)out_bytes, 0x00020000);

    // transposed weight image -> LDS, [dx channel][tap*64 + n], rows padded
#pragma unroll
    for (int p = 0; p < CX * KW / 4 / NTH; ++p) {
        const int f = tid + NTH * p, i = f / (KW / 4), k4 = f - i * (KW / 4);
        *reinterpret_cast<float4*>(&w_l[i * LDW + k4 * 4]) = *reinterpret_cast<const float4*>(a.w + (size_t)i * KW + k4 * 4);
    }

    unsigned x_rel[NX]; int x_rc[NX];
#pragma unroll
    for (int p = 0; p < NX; ++p) {
        const int f = tid + NTH * p, hp = f >> 4, c4 = f & 15;
        const int hr = hp / HC, hc = hp - hr * HC;
        x_rc[p] = ((f < NXF ? hr : 1 << 12) << 8) | hc;
        x_rel[p] = (unsigned)((hr * a.Win + hc) * a.Cin + c4 * 4) * 4u;
    }
    u32x4 rx[NX];
    auto issue = [&](int u) {
        const int uc = u % units_w, t = u / units_w, ur = t % units_h, b = t / units_h;
        const int a0 = ur * UR, c0 = uc * UC;
        const unsigned base = (unsigned)(((b * a.Hin + a0) * a.Win + c0) * a.Cin) * 4u;
        const int lim_r = a.Hin - a0, lim_c = a.Win - c0;                    // dy beyond the lattice is zero
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const bool ok = ((x_rc[p] >> 8) < lim_r) & ((x_rc[p] & 255) < lim_c);
            rx[p] = __builtin_amdgcn_raw_buffer_load_b128(dr, ok ? base + x_rel[p] : OOB, 0, 0);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const int f = tid + NTH * p;
            if (NTH * p + NTH <= NXF || f < NXF) *reinterpret_cast<u32x4*>(&d_l[(f >> 4) * LDD + (f & 15) * 4]) = rx[p];
        }
    };

    // lane -> lattice pixel (la, lb) of the wave's tile for the operand fragments; -> dx channel l31 for the weights and the output
    const int la = rt * 2 + (l31 >> 4), lb = l31 & 15;
    const float* pa = d_l + (la * HC + lb) * LDD + half * 4;
    const float* pb = w_l + l31 * LDW + half * 4;
    // output pixel of accumulator register r: lattice row rt*2 + (r >> 3), lattice column 8 ((r >> 2) & 1) + (r & 3) + 4 half
    const int orow = rt * 2, ocol = 4 * half;
    float b_sc = 0.f, b_sh = 0.f, b_mu = 0.f, b_is = 0.f;
    if (bnred) { b_sc = a.bn_scale[l31]; b_sh = a.bn_shift[l31]; b_mu = a.bn_mean[l31]; b_is = a.bn_invstd[l31]; }
    double db_d = 0.0, dg_d = 0.0;

    // one parity class of the wave's tile: taps (dh, dw, wslot) t = 0..NT-1 in the tile kernel's order
    auto run_class = [&](int ph, int pw, auto ntaps, const int (&tdh)[4], const int (&tdw)[4], const int (&tws)[4], unsigned unit_out,
                         int lim_r, int lim_c) {
        constexpr int NT = decltype(ntaps)::value;
        const unsigned cls_out = unit_out + (unsigned)((ph * a.Wout + pw) * CX) * 4u;
        unsigned zoff[16];
        u32x4 zv[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r >> 3, col = 8 * ((r >> 2) & 1) + (r & 3);
            const bool ok = (row < lim_r) & (col < lim_c);
            zoff[r] = ok ? cls_out + (unsigned)((2 * row * a.Wout + 2 * col) * CX) * 4u : OOB;
        }
        if (bnred) {
#pragma unroll
            for (int r = 0; r < 16; ++r) zv[r >> 2][r & 3] = __builtin_amdgcn_raw_buffer_load_b32(zr, zoff[r], 0, 0);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        auto frag = [&](int i, float4& fa, float4& fb) {          // i = tap * 8 + g
            const int t = i >> 3, g = i & 7;
            fa = *reinterpret_cast<const float4*>(pa + (tdh[t] * HC + tdw[t]) * LDD + g * 8);
            fb = *reinterpret_cast<const float4*>(pb + tws[t] * CD + g * 8);
        };
        float4 fa0, fb0, fa1, fb1;
        frag(0, fa0, fb0);
#pragma unroll
        for (int i = 0; i < NT * 8; i += 2) {
            frag(i + 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.x, fb0.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.y, fb0.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.z, fb0.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.w, fb0.w, acc, 0, 0, 0);
            if (i + 2 < NT * 8) frag(i + 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.x, fb1.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.y, fb1.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.z, fb1.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.w, fb1.w, acc, 0, 0, 0);
        }
        float db = 0.f, dg = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), orr, zoff[r], 0, 0);
            if (bnred) {
                const float z = __uint_as_float(zv[r >> 2][r & 3]);
                const float g = zoff[r] != OOB ? acc[r] : 0.0f;
                const float gy = (z * b_sc + b_sh) > 0.f ? g : g * a.bn_leaky;
                db += gy; dg += gy * ((z - b_mu) * b_is);
            }
        }
        if (bnred) {
            db += __shfl_xor(db, 32); dg += __shfl_xor(dg, 32);
            db_d += (double)db; dg_d += (double)dg;
        }
    };

    issue(u_begin);
    stage();
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
        const bool more = u + 1 < u_end;
        if (more) issue(u + 1);
        const int uc = u % units_w, t = u / units_w, ur = t % units_h, b = t / units_h;
        const int a0 = ur * UR, c0 = uc * UC;
        const unsigned unit_out = (unsigned)(((b * a.Hout + 2 * (a0 + orow)) * a.Wout + 2 * (c0 + ocol)) * CX + l31) * 4u;
        const int lim_r = a.Hin - a0 - orow, lim_c = a.Win - c0 - ocol;
        // taps of a class in the tile kernel's order (ops.hip fv_op_conv_dgrad): r ascending, q ascending among the valid ones;
        // dh = (ph + 1 - r) / 2, dw = (pw + 1 - q) / 2, weight slot r * 3 + q
        if (grp == 0) {
            const int dh[4] = {1, 1, 0, 0}, dw[4] = {1, 0, 1, 0}, ws[4] = {0, 2, 6, 8};               // (ph, pw) = (1, 1)
            run_class(1, 1, std::integral_constant<int, 4>{}, dh, dw, ws, unit_out, lim_r, lim_c);
        } else {
            { const int dh[4] = {1, 0, 0, 0}, dw[4] = {0, 0, 0, 0}, ws[4] = {1, 7, 0, 0};             // (1, 0): r in {0, 2}, q = 1
              run_class(1, 0, std::integral_constant<int, 2>{}, dh, dw, ws, unit_out, lim_r, lim_c); }
            { const int dh[4] = {0, 0, 0, 0}, dw[4] = {1, 0, 0, 0}, ws[4] = {3, 5, 0, 0};             // (0, 1): r = 1, q in {0, 2}
              run_class(0, 1, std::integral_constant<int, 2>{}, dh, dw, ws, unit_out, lim_r, lim_c); }
            { const int dh[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0}, ws[4] = {4, 0, 0, 0};             // (0, 0): r = q = 1
              run_class(0, 0, std::integral_constant<int, 1>{}, dh, dw, ws, unit_out, lim_r, lim_c); }
        }
        __syncthreads();                 // every wave is done reading this unit's halo
        if (more) stage();
        __syncthreads();
    }

    if (bnred) {
        double* red = reinterpret_cast<double*>(d_l);          // [2][8 waves][32]
        if (half == 0) { red[wave * 32 + l31] = db_d; red[(8 + wave) * 32 + l31] = dg_d; }
        __syncthreads();
        if (tid < CX) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += red[k * 32 + tid]; q += red[(8 + k) * 32 + tid]; }
            double* sl = a.bn_slots + (size_t)(blockIdx.x % a.bn_nslot) * 2 * CX;
            unsafeAtomicAdd(sl + tid, s);
            unsafeAtomicAdd(sl + CX + tid, q);
        }
    }
}

}  // namespace

bool fv_dgrad9s2_ok(const FvConvArgs& a) {
    if (a.Cin != CD || a.Nout != CX || a.Tw != 9 || a.nclass != 4 || a.ksplit > 1) return false;
    if (a.epi & ~FV_EPI_BNRED) return false;
    if ((a.epi & FV_EPI_BNRED) && (!a.bn_z || !a.bn_slots || a.bn_nslot < 1)) return false;
    if (a.is != 1 || a.os != 2 || a.Hl != a.Hin || a.Wl != a.Win || a.Hout != 2 * a.Hin || a.Wout != 2 * a.Win) return false;
    // masked dx stores and z loads use the 32-bit byte offset 0x80000000 as "outside": it must lie beyond the output tensor
    if ((long long)a.B * a.Hout * a.Wout * a.Nout * 4 >= (1ll << 31)) return false;
    // the class table fv_op_conv_dgrad builds for a 3x3 stride-2 layer: class c = 3 - (ph * 2 + pw)
    static const int n_exp[4] = {4, 2, 2, 1};
    for (int c = 0; c < 4; ++c) {
        const int ph = (3 - c) >> 1, pw = (3 - c) & 1;
        if (a.oph[c] != ph || a.opw[c] != pw || a.taps[c].n != n_exp[c]) return false;
        int k = 0;
        for (int r = 0; r < 3; ++r) {
            if ((ph + 1 - r) % 2 != 0) continue;
            for (int q = 0; q < 3; ++q) {
                if ((pw + 1 - q) % 2 != 0) continue;
                if (a.taps[c].dh[k] != (ph + 1 - r) / 2 || a.taps[c].dw[k] != (pw + 1 - q) / 2 || a.taps[c].wslot[k] != r * 3 + q) return false;
                ++k;
            }
        }
    }
    return true;
}

int fv_dgrad9s2_launch(fv_ctx* ctx, const FvConvArgs& a) {
    const int units_w = (a.Win + UC - 1) / UC, units_h = (a.Hin + UR - 1) / UR;
    const long long n_units = (long long)a.B * units_h * units_w;
    FV_REQUIRE(ctx, n_units < (1ll << 30), "dgrad9s2: too many units");
    const int grid = n_units < 256 ? (int)n_units : 256;   // one workgroup per CU (116 KB of LDS), contiguous unit ranges
    FvProfScope ps(ctx, "dgrad9s2_kernel", a.alg_flops,
                   4.0 * ((double)a.B * a.Hin * a.Win * a.Cin + (double)a.Nout * a.Tw * a.Cin +
                          (double)a.B * a.Hout * a.Wout * a.Nout * ((a.epi & FV_EPI_BNRED) ? 2 : 1)));
    hipLaunchKernelGGL(dgrad9s2_kernel, dim3(grid), dim3(NTH), 0, ctx->stream, a, units_w, units_h, (int)n_units);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
