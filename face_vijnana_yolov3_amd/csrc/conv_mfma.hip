// fp32 implicit-GEMM gather-convolution on CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Serves the Darknet-53 forward convs (reference yolov3_detect.py:196-215 `_conv_block`:
// ZeroPadding2D(1) + Conv2D 'valid'), the head conv (face_detection.py:348-352) and, with
// mirrored / parity-class tap lists, every data-gradient.  Exact fp32: the MFMA result is a
// k-ordered fmaf chain, so the numerics equal a scalar fp32 convolution.
//
// GEMM view: M = output pixels of the lattice, N = output channels, K = taps x Cin.
//  * block = 256 threads (4 waves), tile 128 x BN (BN = 128/64/32), K step 32
//  * A (pixels x channels) and B (out-channels x channels) tiles are both K-contiguous in HBM
//    (NHWC activations, OHWI weights): 16-byte loads, register-staged into a double-buffered
//    LDS image [row][32+4] whose 36-dword row stride makes the ds_read_b128 fragment reads
//    conflict-free (16 lanes of a group hit 16 distinct 4-bank slots)
//  * K permutation: within an 8-deep chunk lane-half h supplies k = 4h+j to MFMA j, so one
//    ds_read_b128 per operand block feeds four MFMAs
//  * epilogue: optional per-channel affine (+LeakyReLU, +residual add) for inference, or raw
//    store + deterministic per-tile column sums / sums of squares for training-mode BatchNorm
//  * XCD-aware bijective remap of blockIdx so tiles that share an A panel land on one L2
#include <string>
#include <type_traits>
#include "conv_tile.h"

namespace {

// Phase stamps of every workgroup (tools/build_variant.sh stamps -DFV_CONV_STAMPS; tools/conv_phases.py): kernel entry, first
// operand tile staged, K loop done, last store issued -- wall_clock64 (100 MHz, the same clock on every CU) + the hardware id.
#ifdef FV_CONV_STAMPS
constexpr int STAMP_WGS = 32768;
__device__ unsigned long long g_stamps[STAMP_WGS * 5];
#define FV_STAMP(k)                                                                                             \
    do {                                                                                                        \
        if (threadIdx.x == 0) {                                                                                 \
            const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                 \
            if (wg_ < STAMP_WGS) {                                                                              \
                g_stamps[wg_ * 5 + (k)] = wall_clock64();                                                       \
                if ((k) == 0) {                                                                                 \
                    unsigned hw_, xcc_;                                                                         \
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                           \
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                         \
                    g_stamps[wg_ * 5 + 4] = ((unsigned long long)xcc_ << 32) | hw_;                              \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)
#else
#define FV_STAMP(k) do {} while (0)
#endif

// BM_: rows of the tile.  128 everywhere except the K-split launches of the small-M inference path whose row count leaves a 128-row
// tiling more padding than a 64-row one (fv_conv_bm64: 13x13, 26x26, 52x52 pixels at batch 1): 64 x 128 tiles, 2 x 4 waves of 32 x 32.
template <int BN, int WAVES_M, int WAVES_N, bool GATHER, int BM_ = 128>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_kernel(const FvConvArgs a) {
    constexpr int BM = BM_;                // (shadows the 128 of conv_tile.h inside this kernel)
    // NTH threads: 4 waves (2x2, each 64x64) or 8 waves (2x4, each 64x32: two more waves per SIMD to cover barriers and LDS latency)
    constexpr int NTH = 64 * WAVES_M * WAVES_N;
    constexpr int APT = BM * 8 / NTH;      // A-tile float4 loads per thread
    constexpr int RSTEP = NTH / 8;         // tile rows covered by one pass of the workgroup
    static_assert(!GATHER || NTH == 256, "the gather path is written for 4 waves");
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MB = WTM / 32, NB = WTN / 32;
    constexpr int BL = BN * 8 / NTH;  // B-tile float4 loads per thread
    static_assert((NTH == 256 || NTH == 512) && MB >= 1 && NB >= 1 && BL >= 1, "bad tiling");

    // one block: A double buffer, B double buffer; reused as the BM x BN output tile by the epilogue
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDT];
    static_assert(2 * (BM + BN) * LDT >= BM * BN + 2 * (NTH > 256 ? NTH / 64 : NTH / (BN / 4)) * BN, "operand LDS must hold the output tile + the BN-backward reduction scratch");
    float (*As)[BM * LDT] = reinterpret_cast<float (*)[BM * LDT]>(smem);
    float (*Bs)[BN * LDT] = reinterpret_cast<float (*)[BN * LDT]>(smem + 2 * BM * LDT);
    __shared__ int rowoff[BM];
    __shared__ float red[2][WAVES_M][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int cls = blockIdx.z;
    FV_STAMP(0);
    const FvTaps& taps = a.taps[cls];

    const int NT = (a.Nout + BN - 1) / BN;
    // Tail split (speed only): the first tail_full tiles are computed whole; every remaining tile is cut
    // into tail_f K-slices, one workgroup each, dispatched after the whole tiles.  A slice stores its raw
    // partial tile to tail_slab and conv_tail_fixup_kernel adds the slices in fixed order and applies the
    // epilogue -- so the last, partly filled round of workgroups is divided among all CUs.
    const bool tail_part = a.tail_f > 1 && (int)blockIdx.x >= a.tail_full;
    const int tail_q = tail_part ? (int)blockIdx.x - a.tail_full : 0;
    const int tile = a.tail_f > 1 ? (tail_part ? a.tail_full + tail_q / a.tail_f : xcd_remap(blockIdx.x, a.tail_full))
                                  : xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile / NT, nt = tile - mt * NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int HWl = a.Hl * a.Wl;

    // output row offsets (in elements) for the epilogue, -1 = row outside the problem
    if (tid < BM) {
        int m = m0 + tid, off = -1;
        if (m < a.M) {
            int b = m / HWl, rem = m - b * HWl, oh = rem / a.Wl, ow = rem - oh * a.Wl;
            off = ((b * a.Hout + oh * a.os + a.oph[cls]) * a.Wout + ow * a.os + a.opw[cls]) * a.Nout;
        }
        rowoff[tid] = off;
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto compute = [&](const float* __restrict__ Asm, const float* __restrict__ Bsm) {
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 af[MB], bf[NB];
            const int ko = kc * 8 + (lane >> 5) * 4;
#pragma unroll
            for (int i = 0; i < MB; ++i)
                af[i] = *reinterpret_cast<const float4*>(&Asm[(wm * WTM + i * 32 + (lane & 31)) * LDT + ko]);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                bf[j] = *reinterpret_cast<const float4*>(&Bsm[(wn * WTN + j * 32 + (lane & 31)) * LDT + ko]);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };

    if constexpr (GATHER) {
        // first layer: K = 9*Cin <= 32 gathered element-wise (3x3, pad 1, stride 1), one K step
        const int Kg = 9 * a.Cin;
        {
            const int row = tid >> 1, kb = (tid & 1) * 16;
            const int m = m0 + row;
            int b = 0, oh = 0, ow = 0;
            const bool mv = m < a.M;
            if (mv) { b = m / HWl; int rem = m - b * HWl; oh = rem / a.Wl; ow = rem - oh * a.Wl; }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int k = kb + e;
                float v = 0.0f;
                if (mv && k < Kg) {
                    int tp = k / a.Cin, c = k - tp * a.Cin;
                    int ih = oh + tp / 3 - 1, iw = ow + tp % 3 - 1;
                    if ((unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win)
                        v = a.x[((size_t)(b * a.Hin + ih) * a.Win + iw) * a.Cin + c];
                }
                As[0][row * LDT + k] = v;
            }
#pragma unroll
            for (int p = 0; p < BL; ++p) {
                int r = (tid >> 3) + 32 * p, n = n0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < a.Nout) v = *reinterpret_cast<const float4*>(a.w + (size_t)n * BK + (tid & 7) * 4);
                *reinterpret_cast<float4*>(&Bs[0][r * LDT + (tid & 7) * 4]) = v;
            }
        }
        __syncthreads();
        compute(As[0], Bs[0]);
    } else {
        // Operand rows come through buffer descriptors: an out-of-range offset (row outside the
        // image / the problem) returns zeros in hardware, so the K loop has no per-load branches.
        constexpr unsigned OOB = 0x80000000u;  // >= num_records for every tensor we accept (< 2^31 bytes)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.x, 0, (int)((unsigned)a.B * a.Hin * a.Win * a.Cin * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.w, 0, (int)((unsigned)a.Nout * a.Tw * a.Cin * 4u), 0x00020000);
        const int col4 = (tid & 7) * 4;
        int a_pix[APT], a_oh[APT], a_ow[APT];
#pragma unroll
        for (int p = 0; p < APT; ++p) {
            int m = m0 + (tid >> 3) + RSTEP * p;
            if (m < a.M) {
                int b = m / HWl, rem = m - b * HWl, oh = rem / a.Wl, ow = rem - oh * a.Wl;
                a_pix[p] = b * a.Hin; a_oh[p] = oh * a.is; a_ow[p] = ow * a.is;
            } else {
                a_pix[p] = 0; a_oh[p] = -(1 << 28); a_ow[p] = 0;
            }
        }
        unsigned b_row[BL];
#pragma unroll
        for (int p = 0; p < BL; ++p) {
            int n = n0 + (tid >> 3) + RSTEP * p;
            b_row[p] = n < a.Nout ? (unsigned)(n * a.Tw * a.Cin + col4) * 4u : OOB;
        }

        const int cpk = a.Cin / BK;
        const int nk = taps.n * cpk;
        // split-K (small-M inference: blockIdx.y; tail split: K-slice of a tail tile): this block
        // owns K steps [s_begin, s_end)
        const int nslice = tail_part ? a.tail_f : a.ksplit;
        const int per = (nk + nslice - 1) / nslice;
        const int s_begin = (tail_part ? tail_q % a.tail_f : (int)blockIdx.y) * per;
        const int s_end = min(nk, s_begin + per);
        // The K loop is unrolled by two with two register sets -- step s issues the loads of step s + 2 into the set that was
        // staged during step s - 1, and stages the set loaded during step s - 1.  The LDS buffer index becomes a compile-time
        // constant and a staged row has been in its registers for a whole step.  Measured on MI355X (same box, 416x416 batch 40,
        // ms per step of conv_kernel<128,2,4>): one set, rolled loop 28.9; one set, unrolled 28.8; this form 28.2; the same with
        // branch-free (always issued, range-masked) loads and stores, which lets the loads of step s + 2 stay in flight across
        // the staging point, 28.8 -- DESIGN.md 4.1.
        u32x4 ra[APT], rb[BL], ra2[APT], rb2[BL];
        unsigned a_off[APT];
        int t = s_begin / cpk, ci = s_begin - t * cpk;
        auto set_tap = [&](int tp) {
            const int dh = taps.dh[tp], dw = taps.dw[tp];
#pragma unroll
            for (int p = 0; p < APT; ++p) {
                int ih = a_oh[p] + dh, iw = a_ow[p] + dw;
                bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
                a_off[p] = ok ? (unsigned)(((a_pix[p] + ih) * a.Win + iw) * a.Cin + col4) * 4u : OOB;
            }
        };
        auto load = [&]() {
            const int c0b = ci * BK * 4;
            const int wofs = (taps.wslot[t] * a.Cin) * 4 + c0b;
#pragma unroll
            for (int p = 0; p < APT; ++p) ra[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, a_off[p], c0b, 0);
#pragma unroll
            for (int p = 0; p < BL; ++p) rb[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_row[p], wofs, 0);
        };
        auto stage = [&](int buf) {
#pragma unroll
            for (int p = 0; p < APT; ++p)
                *reinterpret_cast<u32x4*>(&As[buf][((tid >> 3) + RSTEP * p) * LDT + col4]) = ra[p];
#pragma unroll
            for (int p = 0; p < BL; ++p)
                *reinterpret_cast<u32x4*>(&Bs[buf][((tid >> 3) + RSTEP * p) * LDT + col4]) = rb[p];
        };
        auto advance = [&]() {
            if (++ci == cpk) { ci = 0; ++t; if (t < taps.n) set_tap(t); }
        };
        auto load2 = [&]() {      // second register set
            const int c0b = ci * BK * 4;
            const int wofs = (taps.wslot[t] * a.Cin) * 4 + c0b;
#pragma unroll
            for (int p = 0; p < APT; ++p) ra2[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, a_off[p], c0b, 0);
#pragma unroll
            for (int p = 0; p < BL; ++p) rb2[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_row[p], wofs, 0);
        };
        auto stage2 = [&](int buf) {
#pragma unroll
            for (int p = 0; p < APT; ++p)
                *reinterpret_cast<u32x4*>(&As[buf][((tid >> 3) + RSTEP * p) * LDT + col4]) = ra2[p];
#pragma unroll
            for (int p = 0; p < BL; ++p)
                *reinterpret_cast<u32x4*>(&Bs[buf][((tid >> 3) + RSTEP * p) * LDT + col4]) = rb2[p];
        };
        // fragment double-buffering: the LDS reads of K-chunk c+1 are issued before the MFMAs of
        // chunk c, and the next tile is staged into the other LDS buffer while chunks 2-3 compute
        const int arow = (wm * WTM + (lane & 31)) * LDT + (lane >> 5) * 4;
        const int brow = (wn * WTN + (lane & 31)) * LDT + (lane >> 5) * 4;
        auto readfrag = [&](const float* __restrict__ Asm, const float* __restrict__ Bsm, int kc, float4 (&af)[MB], float4 (&bf)[NB]) {
#pragma unroll
            for (int i = 0; i < MB; ++i) af[i] = *reinterpret_cast<const float4*>(&Asm[arow + i * 32 * LDT + kc * 8]);
#pragma unroll
            for (int j = 0; j < NB; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bsm[brow + j * 32 * LDT + kc * 8]);
        };
        // k-major order: consecutive MFMAs rotate over all MB*NB accumulators, so an accumulator is
        // re-used only every MB*NB-th instruction (dependent-accumulator latency never on the issue path)
        auto mfma_chunk = [&](const float4 (&af)[MB], const float4 (&bf)[NB]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const float av = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
                        const float bv = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        };

        {
            if (s_begin < s_end) {
                set_tap(t);
                load(); advance();
                if (s_begin + 1 < s_end) { load2(); advance(); }
                stage(0);
            }
            __syncthreads();
            FV_STAMP(1);
            auto body = [&](int s, auto odd) {
                constexpr bool ODD = decltype(odd)::value;      // even steps (from s_begin): LDS 0, next staged from set 2
                const float* Ac = As[ODD ? 1 : 0]; const float* Bc = Bs[ODD ? 1 : 0];
                if (s + 2 < s_end) { if constexpr (ODD) load2(); else load(); advance(); }
                float4 af0[MB], bf0[NB], af1[MB], bf1[NB];
                readfrag(Ac, Bc, 0, af0, bf0);
                readfrag(Ac, Bc, 1, af1, bf1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_chunk(af0, bf0);
                __builtin_amdgcn_sched_barrier(0);
                readfrag(Ac, Bc, 2, af0, bf0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_chunk(af1, bf1);
                __builtin_amdgcn_sched_barrier(0);
                readfrag(Ac, Bc, 3, af1, bf1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_chunk(af0, bf0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < s_end) { if constexpr (ODD) stage(0); else stage2(1); }
                __builtin_amdgcn_sched_barrier(0);
                mfma_chunk(af1, bf1);
                __syncthreads();
            };
            for (int s = s_begin; s < s_end; s += 2) {
                body(s, std::false_type{});
                if (s + 1 < s_end) body(s + 1, std::true_type{});
            }
        }
    }

    // ------------------------------------------------------------------ epilogue
    FV_STAMP(2);
    const int half = lane >> 5, lc = lane & 31;
    if (tail_part) {
        // K-slice of a tail tile: raw partial tile, tile-local [BM][BN] layout, to its slab
        float* Cs = smem;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BN + wn * WTN + j * 32 + lc] = acc[i][j][r];
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(a.tail_slab + (size_t)tail_q * (BM * BN));
#pragma unroll
        for (int p = 0; p < BM * BN / 4 / NTH; ++p) dst[tid + NTH * p] = reinterpret_cast<const float4*>(Cs)[tid + NTH * p];
        return;
    }
    if (a.epi & FV_EPI_STATS) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { float v = acc[i][j][r]; s += v; q += v * v; }
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (half == 0) { red[0][wm][wn * WTN + j * 32 + lc] = s; red[1][wm][wn * WTN + j * 32 + lc] = q; }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.Nout) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) { s += red[0][w][tid]; q += red[1][w][tid]; }
            stat_store(a, mt, n0 + tid, s, q);
        }
    }
    if ((a.Nout & 3) == 0) {
        // Wide store path.  In the accumulator layout a lane owns one output column and 16 scattered
        // rows, i.e. 64 four-byte stores per lane and tile -- a store-issue-bound tail of ~10 us per
        // tile.  Transpose the tile through the (now free) operand LDS and write whole 16-byte pieces:
        // 4x fewer store instructions, every wave instruction covers two full 512-byte rows.
        float* Cs = smem;
        __syncthreads();         // every wave is done reading the operand tiles
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BN + wn * WTN + j * 32 + lc] = acc[i][j][r];
        __syncthreads();
        constexpr int C4 = BN / 4;                   // float4 pieces per tile row
        float* outp = a.out + (size_t)blockIdx.y * a.split_stride;
        const bool bnred = (a.epi & FV_EPI_BNRED) != 0;
        BnRedAcc br;                                 // NTH % C4 == 0: a thread keeps its 4 columns over the rows
        br.init(a, n0 + (tid % C4) * 4, bnred && n0 + (tid % C4) * 4 < a.Nout);
        // The tile leaves in groups of four pieces per thread: the global loads of a group (the residual addend; z of the fused
        // BN-backward reduction) are ALL issued before the first of them is used.  Piece by piece -- load, wait, combine, store --
        // every one of the BM * C4 / NTH pieces paid a full memory round trip (eight s_waitcnt vmcnt(0) in a row): ~10 us per tile,
        // which is most of a 1x1 data-gradient tile's life (4 - 8 K steps).  Rows outside the problem load from row 0 (in range,
        // unused).
        constexpr int NP = BM * C4 / NTH, GP = NP < 4 ? NP : 4;
        static_assert(NP % GP == 0, "epilogue grouping");
        const bool addon = (a.epi & FV_EPI_ADD) != 0;
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += GP) {
            int offn[GP]; bool okp[GP];
            float4 zq[GP], aq[GP];
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const int f = tid + NTH * (p0 + q), row = f / C4, c4 = (f % C4) * 4;
                const int off = rowoff[row], n = n0 + c4;
                okp[q] = off >= 0 && n < a.Nout;
                offn[q] = okp[q] ? off + n : 0;
            }
            if (bnred) {
#pragma unroll
                for (int q = 0; q < GP; ++q) zq[q] = *reinterpret_cast<const float4*>(a.bn_z + offn[q]);
            }
            if (addon) {
#pragma unroll
                for (int q = 0; q < GP; ++q) aq[q] = *reinterpret_cast<const float4*>(a.addend + offn[q]);
            }
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const int f = tid + NTH * (p0 + q), row = f / C4, c4 = (f % C4) * 4;
                const int n = n0 + c4;
                if (okp[q]) {
                    float4 v = *reinterpret_cast<const float4*>(&Cs[row * BN + c4]);
                    if (a.epi & FV_EPI_AFFINE) {
                        if (a.scale) { const float4 s = *reinterpret_cast<const float4*>(a.scale + n); v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
                        if (a.shift) { const float4 s = *reinterpret_cast<const float4*>(a.shift + n); v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
                    }
                    if (a.epi & FV_EPI_LEAKY) {
                        v.x = v.x > 0.f ? v.x : v.x * a.leaky; v.y = v.y > 0.f ? v.y : v.y * a.leaky;
                        v.z = v.z > 0.f ? v.z : v.z * a.leaky; v.w = v.w > 0.f ? v.w : v.w * a.leaky;
                    }
                    if (addon) { v.x += aq[q].x; v.y += aq[q].y; v.z += aq[q].z; v.w += aq[q].w; }
                    *reinterpret_cast<float4*>(outp + offn[q]) = v;
                    if (bnred) br.add(v, zq[q], a.bn_leaky);
                }
            }
        }
        if (bnred) bnred_flush<BN, NTH>(a, br, smem + BM * BN, n0, mt + cls * (int)(gridDim.x / NT), tid);
        FV_STAMP(3);
        return;
    }
    // scalar path: output rows that are not 16-byte aligned (head: 6 channels; 255-channel detection convs)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * WTN + j * 32 + lc;
        const bool nv = n < a.Nout;
        float sc = 1.0f, sh = 0.0f;
        if ((a.epi & FV_EPI_AFFINE) && nv) {
            if (a.scale) sc = a.scale[n];
            if (a.shift) sh = a.shift[n];
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int off = rowoff[row];
                if (off >= 0 && nv) {
                    float v = acc[i][j][r];
                    if (a.epi & FV_EPI_AFFINE) v = v * sc + sh;
                    if (a.epi & FV_EPI_LEAKY) v = v > 0.0f ? v : v * a.leaky;
                    if (a.epi & FV_EPI_ADD) v += a.addend[off + n];
                    a.out[(size_t)blockIdx.y * a.split_stride + off + n] = v;
                }
            }
    }
}

// Fix-up of the tail split: one workgroup per tail tile adds its tail_f raw K-slices in slice order
// (deterministic), then does what the conv epilogue would have done: per-tile column sums / sums of
// squares (training BN) and the affine / LeakyReLU / residual store.  HBM-bound, (tail_f + 1) tiles of
// traffic per tail tile.
template <int BN>
__global__ __launch_bounds__(1024) void conv_tail_fixup_kernel(const FvConvArgs a) {
    constexpr int NTH = 1024;
    constexpr int C4 = BN / 4;                 // float4 pieces per tile row; NTH % C4 == 0, so a thread keeps its columns
    constexpr int RL = NTH / C4;               // rows per pass (32 for BN = 128)
    static_assert(NTH % C4 == 0 && (BM * C4) % NTH == 0, "fix-up tiling");
    __shared__ int rowoff[BM];
    __shared__ float red[2][RL][BN];
    const int tid = threadIdx.x;
    const int NT = (a.Nout + BN - 1) / BN;
    const int tile = a.tail_full + blockIdx.x;
    const int mt = tile / NT, nt = tile - mt * NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int HWl = a.Hl * a.Wl;
    if (tid < BM) {
        int m = m0 + tid, off = -1;
        if (m < a.M) {
            int b = m / HWl, rem = m - b * HWl, oh = rem / a.Wl, ow = rem - oh * a.Wl;
            off = ((b * a.Hout + oh * a.os + a.oph[0]) * a.Wout + ow * a.os + a.opw[0]) * a.Nout;
        }
        rowoff[tid] = off;
    }
    __syncthreads();
    const int c4 = (tid % C4) * 4, rl = tid / C4;
    const int n = n0 + c4;
    const float4* slab = reinterpret_cast<const float4*>(a.tail_slab + (size_t)blockIdx.x * a.tail_f * (BM * BN));
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;
    const bool bnred = (a.epi & FV_EPI_BNRED) != 0;
    BnRedAcc br;
    br.init(a, n, bnred && n < a.Nout);
#pragma unroll
    for (int p = 0; p < BM * C4 / NTH; ++p) {
        const int f = tid + NTH * p, row = f / C4;
        float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bnred && rowoff[row] >= 0 && n < a.Nout) zv = *reinterpret_cast<const float4*>(a.bn_z + rowoff[row] + n);
        float4 v = slab[f];
        for (int k = 1; k < a.tail_f; ++k) {
            const float4 u = slab[(size_t)k * (BM * BN / 4) + f];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
        const int off = rowoff[row];
        if (off >= 0 && n < a.Nout) {
            if (a.epi & FV_EPI_AFFINE) {
                if (a.scale) { const float4 s = *reinterpret_cast<const float4*>(a.scale + n); v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
                if (a.shift) { const float4 s = *reinterpret_cast<const float4*>(a.shift + n); v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
            }
            if (a.epi & FV_EPI_LEAKY) {
                v.x = v.x > 0.f ? v.x : v.x * a.leaky; v.y = v.y > 0.f ? v.y : v.y * a.leaky;
                v.z = v.z > 0.f ? v.z : v.z * a.leaky; v.w = v.w > 0.f ? v.w : v.w * a.leaky;
            }
            if (a.epi & FV_EPI_ADD) {
                const float4 s = *reinterpret_cast<const float4*>(a.addend + off + n);
                v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
            }
            *reinterpret_cast<float4*>(a.out + off + n) = v;
            if (bnred) br.add(v, zv, a.bn_leaky);
        }
    }
    if (bnred) bnred_flush<BN, NTH>(a, br, &red[0][0][0], n0, mt, tid);
    if (a.epi & FV_EPI_STATS) {
        // rows outside the problem are zero in every slice, so they add nothing
        *reinterpret_cast<float4*>(&red[0][rl][c4]) = cs;
        *reinterpret_cast<float4*>(&red[1][rl][c4]) = cq;
        __syncthreads();
        if (tid < BN && n0 + tid < a.Nout) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < RL; ++w) { s += red[0][w][tid]; q += red[1][w][tid]; }
            stat_store(a, mt, n0 + tid, s, q);
        }
    }
}

template <int BN, int WM_, int WN_, bool G, int BM_ = 128>
int launch_cfg(fv_ctx* ctx, const FvConvArgs& a) {
    constexpr int BM = BM_;
    constexpr int NTH = 64 * WM_ * WN_;
    const int MT = (a.M + BM - 1) / BM, NT = (a.Nout + BN - 1) / BN;
    dim3 grid(MT * NT, a.ksplit > 1 ? a.ksplit : 1, a.nclass);
    static const std::string name_s = "conv_kernel<" + std::to_string(BN) + "," + std::to_string(WM_) + "," + std::to_string(WN_) +
                                      (G ? ",true" : ",false") + (BM_ == 128 ? ">" : "," + std::to_string(BM_) + ">");   // rocprofv3's name
    static const char* name = name_s.c_str();
    FvProfScope ps(ctx, name, "M" + std::to_string(a.M) + " N" + std::to_string(a.Nout) + " K" + std::to_string(a.taps[0].n * a.Cin) +
                                  (a.nclass > 1 ? " s2" : "") + (a.ksplit > 1 ? " ks" + std::to_string(a.ksplit) : "") + ((a.epi & FV_EPI_BNRED) ? " r" : ""),
                   a.alg_flops,
                   4.0 * ((double)a.B * a.Hin * a.Win * a.Cin + (double)a.Nout * a.Tw * a.Cin +
                          (double)a.M * a.nclass * a.Nout * ((a.epi & FV_EPI_ADD) ? 2 : 1)));
    FvConvArgs b = a;
    if (b.ksplit < 1) b.ksplit = 1;
    b.tail_f = 1; b.tail_full = 0; b.tail_slab = nullptr;
    if constexpr (!G && BN == 128 && BM_ == 128) {
        // tail split: only with caller scratch (network-level calls), whole-lattice launches, 16-byte rows
        if (ctx->tail_split && ctx->tail_slab && b.ksplit == 1 && a.nclass == 1 && (a.Nout & 3) == 0) {
            int tf = 1, full = 0; long long need = 0;
            fv_conv_tail_plan(a.M, a.Nout, a.taps[0].n * (a.Cin / BK), &tf, &full, &need);
            if (tf > 1 && need <= ctx->tail_slab_floats) {
                b.tail_f = tf; b.tail_full = full; b.tail_slab = ctx->tail_slab;
                const int R = MT * NT - full;
                hipLaunchKernelGGL((conv_kernel<BN, WM_, WN_, G>), dim3(full + R * tf, 1, 1), dim3(NTH), 0, ctx->stream, b);
                FV_LAUNCH_CHECK(ctx);
                hipLaunchKernelGGL((conv_tail_fixup_kernel<BN>), dim3(R), dim3(1024), 0, ctx->stream, b);
                FV_LAUNCH_CHECK(ctx);
                return FV_OK;
            }
        }
    }
    hipLaunchKernelGGL((conv_kernel<BN, WM_, WN_, G, BM_>), grid, dim3(NTH), 0, ctx->stream, b);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

}  // namespace

#ifdef FV_CONV_STAMPS
extern "C" int fv_debug_conv_stamps(fv_ctx* ctx, unsigned long long* out, int nwg) {
    if (!ctx || !out || nwg < 1 || nwg > STAMP_WGS) return FV_ERR_INVALID;
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FV_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)nwg * 5 * sizeof(unsigned long long)));
    return FV_OK;
}
#endif

int fv_conv_mtiles(int M, int Nout) { (void)Nout; return (M + BM - 1) / BM; }

bool fv_conv_narrow(int M, int Nout, int ksteps) {
    if (Nout <= 64 || (Nout & 31) || ksteps < 8 || ksteps > 16) return false;
    const int mt = (M + BM - 1) / BM;
    return mt * ((Nout + 127) / 128) < 64 && mt * (Nout / 32) >= 48;
}

// Small-M inference, 128-wide tiles: 64-row tiles when the launch is in the small-M regime (fewer than 192 tiles of 128 rows, at
// least 8 K steps) and they leave fewer padded rows than 128-row ones (169 rows: 192 instead of 256; 676: 704 / 768; 2704: 2752 /
// 2816) -- padded rows are multiplied like real ones, and the finer tiling needs fewer K slices (fewer slabs for the finish kernel).
bool fv_conv_bm64(int M, int Nout, int ksteps) {
    if (Nout <= 64 || ksteps < 8 || fv_conv_narrow(M, Nout, ksteps)) return false;
    if (((M + BM - 1) / BM) * ((Nout + 127) / 128) >= 192) return false;
    return (M + 63) / 64 * 64 < (M + 127) / 128 * 128;
}

int fv_conv_choose_ksplit(int M, int Nout, int ksteps, bool allow_bm64) {
    if (fv_conv_narrow(M, Nout, ksteps)) return 1;
    const int bn = Nout > 64 ? 128 : (Nout > 32 ? 64 : 32);
    const int bm = allow_bm64 && fv_conv_bm64(M, Nout, ksteps) ? 64 : BM;
    const int tiles = ((M + bm - 1) / bm) * ((Nout + bn - 1) / bn);
    if (tiles >= 192 || ksteps < 8) return 1;         // enough tiles to fill 256 CUs, or nothing to split
    // As many K slices as fit ONE round of the 512 resident workgroup slots (floor, not ceil: 86 tiles x 6 slices = 516 workgroups are
    // two rounds -- tools/bs1_shapes.py, round 5: 29.1 us against 27.0 with 5 slices; 44 x 12 = 528: 29.2 against 24.8 with 10), at
    // least four K steps per slice, no empty slice.
    const int want = 512 / tiles, cap = ksteps / 4;
    int ks = want < cap ? want : cap;
    if (ks < 1) ks = 1;
    const int per = (ksteps + ks - 1) / ks;
    return (ksteps + per - 1) / per;
}

void fv_conv_tail_plan(int M, int Nout, int ksteps, int* tail_f, int* tail_full, long long* slab_floats) {
    *tail_f = 1; *tail_full = 0; *slab_floats = 0;
    if (Nout <= 64 || (Nout & 3)) return;                        // 128-wide tiles only
    const int T = ((M + BM - 1) / BM) * ((Nout + 127) / 128);
    const int slots = 512;                                       // 2 workgroups x 256 CUs
    const int full = (T / slots) * slots, R = T - full;
    if (R == 0) return;
    // Cost model in microseconds, fitted to per-layer timings on MI355X (tools/layer_bench.py): a K step
    // costs 3.8 us when two workgroups share a CU and 2.25 us when a workgroup has the CU to itself, a
    // round of workgroups carries ~8 us of prologue + epilogue; the fix-up kernel is one more launch
    // (~17 us with its gap) and moves (f + 1) tiles of 64 KiB per tail tile.
    auto round_us = [](int nwg, int steps) { return (nwg <= 256 ? 2.25 : 3.8) * steps + 8.0; };
    const double unsplit = round_us(R, ksteps);
    // (A) every tile of the last round cut into f slices
    double best = unsplit;
    int best_f = 1;
    for (int f = 2; f <= 8 && ksteps / f >= 4; ++f) {
        const int per = (ksteps + f - 1) / f, P = R * f;
        if ((f - 1) * per >= ksteps) continue;                   // would leave an empty slice
        double c = (P / slots) * round_us(slots, per) + (P % slots ? round_us(P % slots, per) : 0.0);
        c += 17.0 + 15.0 + (double)R * (f + 1) * 65536.0 / 4.5e6;
        if (c < best) { best = c; best_f = f; }
    }
    const bool a_ok = best_f > 1 && unsplit - best >= 10.0 && best <= 0.95 * unsplit;
    // (B) a last round of 257 ... 511 tiles leaves some CUs with one workgroup, which then runs 1.7x as fast and idles for
    // the rest of the round.  Keep 256 of those tiles whole -- one per CU -- and cut the others into f slices that fill the
    // second slot of every CU in turn: a CU's whole tile shares the pipe with n = ceil(slices / 256) slices one after the
    // other and has it to itself afterwards.  Measured (tools/layer_bench.py, batch 40): the 424-tile launches (13x13 forward,
    // 26x26 data-gradient, 144 K steps) 0.531 -> 0.450 ms = 120 -> 142 TF, the 845- / 848-tile ones 0.476 -> 0.451 and
    // 0.463 -> 0.450 ms; the model below is conservative (it predicts 0.538 for the first), so (B) is taken whenever it predicts
    // any gain, also where (A) qualifies: 53.25 -> 52.5 ms per training step.
    double bestb = unsplit; int fb = 1;
    if (R > 256) {
        const int Rb = R - 256;
        for (int f = 2; f <= 8 && ksteps / f >= 4; ++f) {
            const int per = (ksteps + f - 1) / f, P = Rb * f, n = (P + 255) / 256;
            if ((f - 1) * per >= ksteps || n * per > ksteps) continue;
            double c = n * (per * 3.8 + 8.0) + (ksteps - n * per) * 2.25 + 8.0;
            c += 17.0 + 15.0 + (double)Rb * (f + 1) * 65536.0 / 4.5e6;
            if (c < bestb) { bestb = c; fb = f; }
        }
    }
    const bool b_ok = fb > 1 && unsplit - bestb >= 5.0;
    if (b_ok) {
        *tail_f = fb; *tail_full = full + 256; *slab_floats = (long long)(R - 256) * fb * BM * 128;
    } else if (a_ok) {
        *tail_f = best_f; *tail_full = full; *slab_floats = (long long)R * best_f * BM * 128;
    }
}

static inline bool gather_cin(int cin) { return cin % BK != 0; }

int fv_conv_launch(fv_ctx* ctx, const FvConvArgs& a) {
    FV_REQUIRE(ctx, a.x && a.w && a.out, "conv: NULL tensor");
    FV_REQUIRE(ctx, a.M > 0 && a.Nout > 0 && a.nclass >= 1 && a.nclass <= 4, "conv: bad problem size");
    FV_REQUIRE(ctx, (long long)a.B * a.Hin * a.Win * a.Cin < (1ll << 29) &&
                        (long long)a.B * a.Hout * a.Wout * a.Nout < (1ll << 31) &&
                        (long long)a.Nout * a.Tw * a.Cin < (1ll << 29),
               "conv: input/weight tensor exceeds 2^29 elements (2 GiB buffer descriptor) or output 2^31");
    FV_REQUIRE(ctx, !(a.epi & FV_EPI_STATS) || (((a.psum && a.psq) || (a.stat_slots && a.stat_nslot >= 1)) && a.nclass == 1),
               "conv: stats need psum/psq or accumulator slots");
    FV_REQUIRE(ctx, !(a.epi & FV_EPI_ADD) || a.addend, "conv: FV_EPI_ADD needs addend");
    FV_REQUIRE(ctx, !(a.epi & FV_EPI_BNRED) || (a.bn_z && a.bn_scale && a.bn_shift && a.bn_mean && a.bn_invstd && a.bn_slots &&
                                                 a.bn_nslot >= 1 && (a.Nout & 3) == 0 && a.ksplit <= 1 &&
                                                 !(a.epi & (FV_EPI_STATS | FV_EPI_AFFINE | FV_EPI_LEAKY))),
               "conv: fused BN-backward reduction needs its layer's tensors, 16-byte rows and a plain (+add) epilogue");
    FV_REQUIRE(ctx, a.ksplit <= 1 || (a.epi == 0 && a.nclass == 1 && a.Cin % BK == 0), "conv: split-K stores raw partials only");
    if (ctx->conv_halo && fv_conv9_fwd_ok(a)) return fv_conv9_fwd_launch(ctx, a);
    if (ctx->conv_halo && fv_dgrad9s2_ok(a)) return fv_dgrad9s2_launch(ctx, a);
    const bool gather = a.Cin % BK != 0;
    if (gather) {
        FV_REQUIRE(ctx, 9 * a.Cin <= BK && a.nclass == 1 && a.is == 1 && a.os == 1 && a.taps[0].n == 9 &&
                            a.Hl == a.Hin && a.Wl == a.Win,
                   "conv: Cin=%d is only supported as a 3x3 stride-1 pad-1 layer with 9*Cin<=32 "
                   "(weights packed [n][32])", a.Cin);
        if (ctx->conv0_direct && fv_conv0_direct_ok(a)) return fv_conv0_direct_launch(ctx, a);
        if (a.Nout > 64) return launch_cfg<128, 2, 2, true>(ctx, a);
        if (a.Nout > 32) return launch_cfg<64, 2, 2, true>(ctx, a);
        return launch_cfg<32, 4, 1, true>(ctx, a);
    }
    for (int c = 0; c < a.nclass; ++c)
        FV_REQUIRE(ctx, a.taps[c].n >= 1 && a.taps[c].n <= 9, "conv: bad tap count");
    if (a.small) return fv_conv_small_launch(ctx, a);
    if (a.narrow) return launch_cfg<32, 4, 1, false>(ctx, a);
    if (ctx->conv1x1_persist && fv_conv1x1_persist_ok(a)) return fv_conv1x1_persist_launch(ctx, a);
    // 128-wide tiles: 8 waves (2 x 4, each 64 x 32) put four waves on every SIMD instead of two: the same per-element fmaf
    // chain (bit-identical outputs), 129 against 121 TF on the 52x52 layers
    if (a.Nout > 64 && a.bm64) return launch_cfg<128, 2, 4, false, 64>(ctx, a);
    if (a.Nout > 64) return ctx->conv_waves8 ? launch_cfg<128, 2, 4, false>(ctx, a) : launch_cfg<128, 2, 2, false>(ctx, a);
    if (a.Nout > 32) return ctx->conv_waves8 ? launch_cfg<64, 4, 2, false>(ctx, a) : launch_cfg<64, 2, 2, false>(ctx, a);
    return launch_cfg<32, 4, 1, false>(ctx, a);
}
