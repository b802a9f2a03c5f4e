// Small-M inference convolution (batch 1: `self.model.predict(image)` of detect(), face_detection.py:899): the K dimension split
// INSIDE the workgroup.
//
// At batch 1 a Darknet-53 layer is 0.2 - 1.6 GFLOP on 169 ... 2704 output pixels.  The tile kernels reach parallelism by cutting K
// into slices of one workgroup each (conv_mfma.hip, fv_conv_choose_ksplit): ~500 workgroups store 32 - 64 KB partial tiles and a
// finish launch sums 5 - 21 slabs per output (tools/bs1_shapes.py: 23 - 27 us + 7 us per 3x3 layer, and a 1x1 layer either
// pays the same finish or walks 8 - 16 dependent K steps on 48 - 88 workgroups).  Here ONE launch of at most 256 workgroups (one
// per CU) does the layer: a 512-thread workgroup owns a BM x BN output tile and its Q wave groups (four pairs, or eight single
// waves) each multiply 1/Q of the K steps from their own double-buffered operand tiles in LDS -- Q independent load -> LDS -> MFMA
// chains per CU instead of one --, then groups 1 .. Q-1 park their accumulators in LDS and group 0's layout adds them in group
// order (deterministic, whatever finishes first), followed by the inference epilogue (affine, LeakyReLU, residual).  No slabs, no
// finish launch.  Gather addressing as in conv_kernel (taps, stride, out-of-image rows read as zeros through the buffer
// descriptor); forward launches only (dense output lattice).
//   <64, 64, 4>  52x52 3x3 layers (172 workgroups, 9 steps per group)        <64, 32, 4>  26x26 3x3, 52x52 1x1 (176 / 172, 18 / 2)
//   <32, 32, 8>  13x13 3x3 and 1x1, 26x26 1x1 (192 / 96 / 176 workgroups, 18 / 4 / 2 steps)
// fv_conv_small_plan picks the configuration with the shortest chain that fits 256 workgroups, or none (the tile kernels).
#include <string>
#include "conv_tile.h"

namespace {

template <int TM, int TN, int Q, int PF>
__global__ __launch_bounds__(512, 2) void conv_small_kernel(const FvConvArgs a) {
    constexpr int WQ = 8 / Q;                    // waves per K group: each owns 32 rows of the tile
    static_assert(TM == 32 * WQ && (TN == 32 || TN == 64) && (Q == 4 || Q == 8), "bad small-M tiling");
    constexpr int NBK = TN / 32;                 // 32 x 32 accumulator blocks per wave
    // Staging unit = the waves that share one set of operand tiles in LDS.  TN = 32: every WAVE has its own (its 32 A rows + a private
    // copy of the 32 B rows: 18 KB, 147 KB per workgroup) -- nothing in the K loop is shared between waves, so it contains NO barrier
    // and the eight waves run free (the shared form synchronised all eight at every step although only a pair shares tiles: 1.65 us
    // per step for 0.85 us of matrix time).  TN = 64: a K group's two waves share their tiles (a private B copy would not fit) and
    // keep the workgroup barrier per step.
    constexpr bool PRIV = TN == 32;
    constexpr int SW = PRIV ? 1 : WQ;            // waves per staging unit
    constexpr int NU = 8 / SW;                   // staging units per workgroup
    constexpr int AR = 32 * SW;                  // A rows per unit
    constexpr int ST = 64 * SW;                  // threads per unit
    constexpr int RS = ST / 8;                   // tile rows covered by one pass of a unit's threads (8 threads x 16 B = one 32-float row)
    constexpr int APT = AR / RS, BPT = TN / RS;  // float4 loads per thread and K step: 4 and 4
    constexpr int UF = 2 * (AR + TN) * LDT;      // one unit's double-buffered A + B tiles (floats)
    __shared__ __attribute__((aligned(16))) float smem[NU * UF];
    static_assert(NU * UF >= Q * TM * TN, "the operand LDS must hold the parked accumulator tiles and their sum");
    static_assert(NU * UF * 4 <= 160 * 1024, "LDS");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = wave / WQ, wr = wave % WQ;
    const int ut = tid % ST;                     // thread index inside the staging unit
    const int arow0 = PRIV ? wr * 32 : 0;        // tile-local row of the unit's first A row
    const int NT = a.Nout / TN;
    const int mt = blockIdx.x / NT, nt = blockIdx.x - mt * NT;
    const int m0 = mt * TM, n0 = nt * TN;
    const FvTaps& taps = a.taps[0];
    const int cpk = a.Cin / BK;
    const int nk = taps.n * cpk, per = nk / Q;   // K steps per group (nk % Q == 0: fv_conv_small_plan)
    float* As0 = smem + (PRIV ? wave : q) * UF;  // [2][AR][LDT]
    float* Bs0 = As0 + 2 * AR * LDT;             // [2][TN][LDT]

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((unsigned)a.B * a.Hin * a.Win * a.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((unsigned)a.Nout * a.Tw * a.Cin * 4u), 0x00020000);
    const int col4 = (ut & 7) * 4, r0 = ut >> 3;
    const int HWl = a.Hl * a.Wl;
    int a_pix[APT], a_oh[APT], a_ow[APT];        // image row base, input row / column of tap (0, 0) for this thread's A rows
#pragma unroll
    for (int p = 0; p < APT; ++p) {
        const int m = m0 + arow0 + r0 + RS * p;
        if (m < a.M) {
            const int b = m / HWl, rem = m - b * HWl, oh = rem / a.Wl, ow = rem - oh * a.Wl;
            a_pix[p] = b * a.Hin; a_oh[p] = oh * a.is; a_ow[p] = ow * a.is;
        } else {
            a_pix[p] = 0; a_oh[p] = -(1 << 28); a_ow[p] = 0;          // every tap lands outside the image: zeros
        }
    }
    unsigned b_row[BPT];
#pragma unroll
    for (int p = 0; p < BPT; ++p) b_row[p] = (unsigned)((n0 + r0 + RS * p) * a.Tw * a.Cin + col4) * 4u;   // Nout % TN == 0: in range

    u32x4 ra[PF][APT], rb[PF][BPT];
    // Steps are requested in order, so (tap, channel chunk) of the NEXT request advance incrementally and the row offsets of a tap are
    // worked out once per tap, as in conv_kernel (the first version divided and redid the bounds tests in every step: ~100 vector
    // instructions per thread and step, which compete with the MFMAs for the issue slot).
    int lt = (q * per) / cpk, lci = (q * per) - lt * cpk;
    unsigned a_off[APT];
    auto set_tap = [&](int t) {
        const int dh = taps.dh[t], dw = taps.dw[t];
#pragma unroll
        for (int p = 0; p < APT; ++p) {
            const int ih = a_oh[p] + dh, iw = a_ow[p] + dw;
            const bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
            a_off[p] = ok ? (unsigned)(((a_pix[p] + ih) * a.Win + iw) * a.Cin + col4) * 4u : OOB;
        }
    };
    set_tap(lt);
    auto load = [&](int slot) {                  // the next K step of THIS group
        const int c0b = lci * BK * 4;
        const int wofs = taps.wslot[lt] * a.Cin * 4 + c0b;
#pragma unroll
        for (int p = 0; p < APT; ++p) ra[slot][p] = __builtin_amdgcn_raw_buffer_load_b128(xr, a_off[p], c0b, 0);
#pragma unroll
        for (int p = 0; p < BPT; ++p) rb[slot][p] = __builtin_amdgcn_raw_buffer_load_b128(wr_, b_row[p], wofs, 0);
        if (++lci == cpk) { lci = 0; ++lt; if (lt < taps.n) set_tap(lt); }
    };
    auto stage = [&](int buf, int slot) {
#pragma unroll
        for (int p = 0; p < APT; ++p) *reinterpret_cast<u32x4*>(&As0[buf * AR * LDT + (r0 + RS * p) * LDT + col4]) = ra[slot][p];
#pragma unroll
        for (int p = 0; p < BPT; ++p) *reinterpret_cast<u32x4*>(&Bs0[buf * TN * LDT + (r0 + RS * p) * LDT + col4]) = rb[slot][p];
    };
    f32x16 acc[NBK];
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const int arow = ((PRIV ? 0 : wr * 32) + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int brow = (lane & 31) * LDT + (lane >> 5) * 4;
    auto compute = [&](int cur) {                // the chunk / lane-half k order of conv_kernel
        const float* Ac = As0 + cur * AR * LDT; const float* Bc = Bs0 + cur * TN * LDT;
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            const float4 af = *reinterpret_cast<const float4*>(&Ac[arow + kc * 8]);
            float4 bf[NBK];
#pragma unroll
            for (int j = 0; j < NBK; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bc[brow + j * 32 * LDT + kc * 8]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NBK; ++j) {
                    const float av = e == 0 ? af.x : e == 1 ? af.y : e == 2 ? af.z : af.w;
                    const float bv = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                }
        }
    };

    // A ring of PF register slots keeps PF K steps' operand rows in flight: step s is staged from slot s % PF, which is at once
    // refilled with step s + PF (first version: rounds of PF loads, then PF multiplications -- every round waited a full memory
    // latency with nothing to multiply: 1.7 us per step where the matrix pipe needs 0.85).  Double-buffered LDS tiles: step s is
    // staged into buffer s & 1 behind the barrier of step s - 1, which every wave passes only after it has read step s - 2.
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (j < per) load(j);
    for (int s0 = 0; s0 < per; s0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int s = s0 + j;
            if (s < per) {                       // uniform over the workgroup
                stage((PF & 1) ? (s & 1) : (j & 1), j);
                if (s + PF < per) load(j);
                if constexpr (!PRIV) __syncthreads();    // private tiles: the wave's own program order is all the ordering there is to keep
                compute((PF & 1) ? (s & 1) : (j & 1));
            }
        }
    }
    __syncthreads();                             // every wave is done with the operand tiles: the hand-off below reuses them
    // hand-off: groups 1 .. Q-1 park their accumulators tile-local [TM][TN]; group 0's layout adds them in group order
    const int half = lane >> 5, lc = lane & 31;
    float* H = smem;                             // [Q-1][TM][TN] parked tiles, then [TM][TN] the sum
    if (q > 0) {
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                H[(q - 1) * TM * TN + (wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TN + j * 32 + lc] = acc[j][r];
    }
    __syncthreads();
    float* Cs = smem + (Q - 1) * TM * TN;
    if (q == 0) {
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = (wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TN + j * 32 + lc;
                float v = acc[j][r];
#pragma unroll
                for (int g = 0; g < Q - 1; ++g) v += H[g * TM * TN + e];
                Cs[e] = v;
            }
    }
    __syncthreads();
    // epilogue over the dense output lattice: row m of the tile is output pixel m
    constexpr int C4 = TN / 4;
    for (int f = tid; f < TM * C4; f += 512) {
        const int row = f / C4, c4 = (f % C4) * 4;
        const int m = m0 + row, n = n0 + c4;
        if (m >= a.M) continue;
        float4 v = *reinterpret_cast<const float4*>(&Cs[row * TN + c4]);
        if (a.epi & FV_EPI_AFFINE) {
            if (a.scale) { const float4 sc = *reinterpret_cast<const float4*>(a.scale + n); v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
            if (a.shift) { const float4 sh = *reinterpret_cast<const float4*>(a.shift + n); v.x += sh.x; v.y += sh.y; v.z += sh.z; v.w += sh.w; }
        }
        if (a.epi & FV_EPI_LEAKY) {
            v.x = v.x > 0.f ? v.x : v.x * a.leaky; v.y = v.y > 0.f ? v.y : v.y * a.leaky;
            v.z = v.z > 0.f ? v.z : v.z * a.leaky; v.w = v.w > 0.f ? v.w : v.w * a.leaky;
        }
        const size_t off = (size_t)m * a.Nout + n;
        if (a.epi & FV_EPI_ADD) { const float4 sk = *reinterpret_cast<const float4*>(a.addend + off); v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w; }
        *reinterpret_cast<float4*>(a.out + off) = v;
    }
}

struct SmallCfg { int tm, tn, q; double step_us; };
// K step of one group as measured (tools/bs1_shapes.py, round 5; every fitting layer forced through this kernel): 1.4 us with one
// 32 x 32 accumulator block per wave and private tiles (1.65 us when the eight waves met at a barrier every step), 2.9 us with two
// blocks and shared tiles -- against 0.85 / 1.7 us of matrix time (16 / 32 MFMAs per wave, two waves per SIMD): one dependent
// accumulator chain per wave and two waves per SIMD; conv_kernel's K loop (four waves per SIMD from two workgroups) reaches 89 %.
const SmallCfg kCfg[3] = {{64, 64, 4, 2.9}, {64, 32, 4, 1.4}, {32, 32, 8, 1.4}};

template <int TM, int TN, int Q>
void launch_small(fv_ctx* ctx, const FvConvArgs& a, int grid, int per) {
    if (per >= 4) hipLaunchKernelGGL((conv_small_kernel<TM, TN, Q, 4>), dim3(grid), dim3(512), 0, ctx->stream, a);
    else if (per >= 2) hipLaunchKernelGGL((conv_small_kernel<TM, TN, Q, 2>), dim3(grid), dim3(512), 0, ctx->stream, a);
    else hipLaunchKernelGGL((conv_small_kernel<TM, TN, Q, 1>), dim3(grid), dim3(512), 0, ctx->stream, a);
}

}  // namespace

// Which configuration (1 .. 3 = index into kCfg + 1), or 0: the launch stays with the tile kernels.  A configuration fits when K
// divides into its groups, the output channels into its tiles and the launch into one workgroup per CU; the one with the shortest
// chain wins if it beats what the tile kernels take for such a layer: ~13 us for a 1x1 layer (8 - 16 dependent K steps on 128 x 32
// tiles, or K slices + finish), ~28 us for a 3x3 layer (one round of K slices + the finish launch).  In Darknet-53 at batch 1 that
// takes every 1x1 layer from 104 x 104 down (9 - 15 us instead of 12 - 19) and the stride-2 3x3 layers into 26 x 26 and 13 x 13
// (9 steps per group); the stride-1 3x3 layers (18 steps per group: 32 - 33 us measured against 30 - 33 for two launches, and slower in
// the un-instrumented forward: 1.08 against 1.01 ms) stay with the tile kernels.
int fv_conv_small_plan(int M, int Nout, int Cin, int ntaps) {
    if (M < 1 || Cin % BK != 0 || ntaps < 1) return 0;
    const int nk = ntaps * (Cin / BK);
    int best = 0;
    double best_us = ntaps == 1 ? 13.0 : 28.0;
    for (int c = 0; c < 3; ++c) {
        const SmallCfg& k = kCfg[c];
        if (nk % k.q != 0 || Nout % k.tn != 0) continue;
        const long long wgs = (long long)((M + k.tm - 1) / k.tm) * (Nout / k.tn);
        if (wgs > 256) continue;
        const double us = (nk / k.q) * k.step_us + 6.0;
        if (us < best_us) { best_us = us; best = c + 1; }
    }
    return best;
}

int fv_conv_small_launch(fv_ctx* ctx, const FvConvArgs& a) {
    FV_REQUIRE(ctx, a.small >= 1 && a.small <= 3 && a.nclass == 1 && a.os == 1 && a.ksplit <= 1 && a.Hout == a.Hl && a.Wout == a.Wl &&
                        !(a.epi & (FV_EPI_STATS | FV_EPI_BNRED)), "conv_small: forward inference launches only");
    const SmallCfg& k = kCfg[a.small - 1];
    const int nk = a.taps[0].n * (a.Cin / BK), per = nk / k.q;
    FV_REQUIRE(ctx, nk % k.q == 0 && a.Nout % k.tn == 0, "conv_small: configuration does not fit the problem");
    const int grid = ((a.M + k.tm - 1) / k.tm) * (a.Nout / k.tn);
    static const char* names[3] = {"conv_small_kernel<64,64,4>", "conv_small_kernel<64,32,4>", "conv_small_kernel<32,32,8>"};
    FvProfScope ps(ctx, names[a.small - 1], "M" + std::to_string(a.M) + " N" + std::to_string(a.Nout) + " K" + std::to_string(a.taps[0].n * a.Cin), a.alg_flops,
                   4.0 * ((double)a.B * a.Hin * a.Win * a.Cin + (double)a.Nout * a.Tw * a.Cin + (double)a.M * a.Nout * ((a.epi & FV_EPI_ADD) ? 2 : 1)));
    if (a.small == 1) launch_small<64, 64, 4>(ctx, a, grid, per);
    else if (a.small == 2) launch_small<64, 32, 4>(ctx, a, grid, per);
    else launch_small<32, 32, 8>(ctx, a, grid, per);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
