// fp32 weight-gradient of the gather-convolution on CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   dw[n][wslot[t]][c] += sum over lattice pixels m of  dy[m][n] * x[pixel(m, t)][c]
//
// Replaces the conv-kernel gradients TF autodiff produces inside Keras `fit_generator`
// (reference face_detection.py:621-627) for every conv of yolov3_detect.py:221-267 and the head.
//
// GEMM view per tap: M' = output channels n, N' = input channels c, K' = pixels.  Both operand
// tiles are read as pixel rows (channels contiguous in NHWC), staged [pixel][channel] in LDS and
// consumed with conflict-free ds_read_b32 (32 consecutive dwords per lane half).
//  * QUAD  (128x128 tile): 2x2 waves, each a 64x64 sub-tile over all 32 pixels of a chunk
//  * split (32/64 tiles) : every wave owns the whole tile over its quarter of the pixel chunk
//  * grid = tiles x taps x K-splits in one dimension, a K-split pinned to one XCD (its workgroups
//    share the x / dy chunks in that L2); the split count fills whole rounds of the XCD's 64
//    workgroup slots; partial tiles are added with float atomics shaped as two 128-byte row segments
//    per wave instruction; the caller zeroes dw once per step
//  * chunk loop: global loads one chunk ahead, LDS fragments one k-pair ahead (register double buffer)
#include <type_traits>
#include "conv_tile.h"

namespace {

constexpr int KP = 32;  // pixels per chunk

// Phase stamps (tools/build_variant.sh stamps -DFV_CONV_STAMPS; tools/conv_phases.py): entry, first chunk staged, chunk loop done, atomics issued
#ifdef FV_CONV_STAMPS
constexpr int WSTAMP_WGS = 8192;
__device__ unsigned long long g_wstamps[WSTAMP_WGS * 5];
#define FV_WSTAMP(k)                                                                          \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < WSTAMP_WGS) {                                    \
            g_wstamps[blockIdx.x * 5 + (k)] = wall_clock64();                                 \
            if ((k) == 0) {                                                                   \
                unsigned hw_, xcc_;                                                           \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));             \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));           \
                g_wstamps[blockIdx.x * 5 + 4] = ((unsigned long long)xcc_ << 32) | hw_;        \
            }                                                                                 \
        }                                                                                     \
    } while (0)
#else
#define FV_WSTAMP(k) do {} while (0)
#endif

// NW = 8 (QUAD only): 512-thread workgroups, 2 x 4 waves of 64 x 32 -- four waves per SIMD instead of two, same arithmetic.
template <int TM, int TN, bool QUAD, bool GATHER, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void wgrad_kernel(const FvWgradArgs a, int nsplit, int ntap_eff, int tiles_taps, int pinned) {
    static_assert(NW == 4 || (NW == 8 && QUAD && !GATHER), "8 waves: the 128x128 form only");
    constexpr int NTH = 64 * NW;
    constexpr int LDA = TM + 4, LDB = TN + 4;  // +4 keeps 16-byte row alignment for the staged float4 writes
    constexpr int WTM = QUAD ? TM / 2 : TM, WTN = QUAD ? TN / (NW / 2) : TN;
    constexpr int MB = WTM / 32, NB = WTN / 32;
    constexpr int AL = TM * KP / 4 / NTH, BL = TN * KP / 4 / NTH;  // float4 loads per thread (>=1)
    static_assert(AL >= 1 && BL >= 1, "tile too small for the workgroup");

    __shared__ __attribute__((aligned(16))) float As[2][KP * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][KP * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    FV_WSTAMP(0);
    const int wm = QUAD ? wave / (NW / 2) : 0, wn = QUAD ? wave % (NW / 2) : 0;
    const int NTc = GATHER ? 1 : a.Cin / TN;
    // Workgroup ids go round-robin over the 8 XCDs (one L2 each).  All (tile, tap) workgroups of one
    // K-split read the same 32-pixel chunks of x and dy, so a split is pinned to one XCD: id = 8 j + xcd,
    // split = 8 (j / tiles) + xcd -- its blocks then hit in that XCD's L2 instead of each going to the fabric.
    int split, bid;
    if (pinned) {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        split = (jx / tiles_taps) * 8 + xcd;
        bid = jx % tiles_taps;
    } else {   // a split already has >= 64 workgroups: every XCD gets an eighth of each split
        split = blockIdx.x / tiles_taps;
        bid = blockIdx.x - split * tiles_taps;
    }
    const int tp = bid % ntap_eff; bid /= ntap_eff;   // taps fastest
    const int ct = bid % NTc, nt = bid / NTc;
    const int n0 = nt * TM, c0 = ct * TN;
    const int HWl = a.Hl * a.Wl;
    const int dh = GATHER ? 0 : a.taps.dh[tp], dw = GATHER ? 0 : a.taps.dw[tp];

    const int total_chunks = (a.M + KP - 1) / KP;
    // balanced partition of the chunks over the splits (sizes differ by at most one)
    const int ch_begin = (int)((long long)split * total_chunks / nsplit);
    const int ch_end = (int)((long long)(split + 1) * total_chunks / nsplit);
    if (split >= nsplit || ch_begin >= ch_end) return;   // padded split ids (pinned mode) own no chunks

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Operand rows come through buffer descriptors (out-of-range offset -> zeros, no branches).
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.B * a.Hin * a.Win * a.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.M * a.Ndy * 4u), 0x00020000);

    // per-thread rows of the A' (dy) and B' (x) tiles; the pixel coordinates of the B' rows are
    // advanced incrementally by KP pixels per chunk (no integer division in the loop)
    unsigned a_off[AL];   // byte offset of dy row at chunk ch_begin (OOB if the channel group is outside N)
    int a_m[AL];
#pragma unroll
    for (int p = 0; p < AL; ++p) {
        int f = tid + NTH * p, row = f / (TM / 4), col = (f % (TM / 4)) * 4;
        a_m[p] = ch_begin * KP + row;
        a_off[p] = (n0 + col < a.N) ? (unsigned)(n0 + col) * 4u : OOB;
    }
    const int adv_b = KP / HWl, adv_h = (KP - adv_b * HWl) / a.Wl, adv_w = KP - adv_b * HWl - adv_h * a.Wl;
    int g_m = 0, g_oh = 0, g_ow = 0, g_dh[4] = {0, 0, 0, 0}, g_dw[4] = {0, 0, 0, 0}, g_rel[4] = {0, 0, 0, 0};
    bool g_kv[4] = {false, false, false, false};
    if constexpr (GATHER) {
        g_m = ch_begin * KP + (tid >> 3);
        const int b = g_m / HWl, rem = g_m - b * HWl;
        g_oh = rem / a.Wl; g_ow = rem - g_oh * a.Wl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = (tid & 7) * 4 + e, t9 = k / a.Cin, c = k - t9 * a.Cin;
            g_kv[e] = k < 9 * a.Cin;
            g_dh[e] = t9 / 3 - 1; g_dw[e] = t9 % 3 - 1;
            g_rel[e] = (g_dh[e] * a.Win + g_dw[e]) * a.Cin + c;
        }
    }
    // SROW (every form but the gathered first layer).  The 32 x rows of a chunk are 32 consecutive lattice pixels; every lane used to carry the pixel
    // coordinates of its rows and redo the carries, the bounds test and the offset product per chunk (~70 vector instructions per
    // lane and chunk, issued by all waves right after the barrier: the kernel ran 11 % below the same kernel with constant offsets).
    // Now lanes 0-31 of wave 0 own one row each, advance it incrementally (KP pixels = adv_b images + adv_h rows + adv_w pixels; + K1
    // when the column wraps, + K2 when the row wraps), and publish the masked byte offset in a 32-entry LDS table, one chunk before
    // the loads that use it; a load's address is one LDS read plus the lane's column.  dy rows are linear in the pixel index.
    // (Per-row arithmetic on the scalar unit was the first attempt: 60 scalar instructions per wave and chunk on the CU's single
    // scalar pipe recovered 4.5 %.)
    constexpr bool SROW = !GATHER;
    constexpr int RPP = NTH / (TN / 4);                    // rows per pass of the workgroup
    __shared__ unsigned rowtab[2][KP];
    int t_oh = 0, t_ow = 0;
    unsigned t_off = 0, tK0 = 0, tK1 = 0, tK2 = 0, lin_a[AL], lane_col = 0;
    // row table of the chunk `rel` chunks after ch_begin -> rowtab[rel & 1]; then the producer lanes step to the next chunk
    auto publish_rows = [&](int rel) {
        if constexpr (SROW) {
            if (tid < KP) {
                const int ih = t_oh * a.is + dh, iw = t_ow * a.is + dw;
                const bool ok = ((unsigned)ih < (unsigned)a.Hin) & ((unsigned)iw < (unsigned)a.Win);   // pixels >= M: past num_records
                rowtab[rel & 1][tid] = ok ? t_off : OOB;
                t_ow += adv_w;
                const bool c1 = t_ow >= a.Wl;
                t_ow -= c1 ? a.Wl : 0; t_oh += adv_h + (c1 ? 1 : 0);
                const bool c2 = t_oh >= a.Hl;
                t_oh -= c2 ? a.Hl : 0;
                t_off += tK0 + (c1 ? tK1 : 0u) + (c2 ? tK2 : 0u);
            }
        }
    };
    if constexpr (SROW) {
        const unsigned px = (unsigned)(a.is * a.Cin) * 4u, rowst = (unsigned)(a.is * a.Win * a.Cin) * 4u,
                       imgst = (unsigned)(a.Hin * a.Win * a.Cin) * 4u;
        tK0 = (unsigned)adv_b * imgst + (unsigned)adv_h * rowst + (unsigned)adv_w * px;
        tK1 = rowst - (unsigned)a.Wl * px;
        tK2 = imgst - (unsigned)a.Hl * rowst;
        lane_col = (unsigned)(c0 + (tid % (TN / 4)) * 4) * 4u;      // (NTH is a multiple of TN / 4: the same column in every pass)
        if (tid < KP) {
            const int m = ch_begin * KP + tid;
            const int b = m / HWl, rem = m - b * HWl;
            t_oh = rem / a.Wl; t_ow = rem - t_oh * a.Wl;
            // modular: a tap above / left of the first pixel gives a "negative" offset that the bounds test masks
            t_off = (unsigned)(((b * a.Hin + t_oh * a.is + dh) * a.Win + t_ow * a.is + dw) * a.Cin) * 4u;
        }
        publish_rows(0);
        publish_rows(1);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < AL; ++p) lin_a[p] = (unsigned)a_m[p] * (unsigned)a.Ndy * 4u + a_off[p];   // rows >= M: past num_records
    }
    int load_rel = 0;    // chunk (relative to ch_begin) the next load() fetches
    u32x4 ra[AL], rb[BL];
    auto load = [&]() {
#pragma unroll
        for (int p = 0; p < AL; ++p) {
            // a_off carries the OOB bit for channel groups outside N; rows past M get it here (no branches)
            unsigned off;
            if constexpr (SROW) { off = lin_a[p]; lin_a[p] += KP * (unsigned)a.Ndy * 4u; }
            else off = ((unsigned)a_m[p] * (unsigned)a.Ndy * 4u + a_off[p]) | (a_m[p] < a.M ? 0u : OOB);
            ra[p] = __builtin_amdgcn_raw_buffer_load_b128(yr, off, 0, 0);
            a_m[p] += KP;
        }
        if constexpr (GATHER) {
            // row = pixel, 32 k-slots = 9 taps x Cin (3x3, pad 1, stride 1), zero padded.  The lane's four k-slots never change
            // (g_dh / g_dw / g_rel, set up once); its pixel advances by KP per chunk with the branch-free carries, and with
            // Hl == Hin, Wl == Win the element index of (pixel, tap, channel) is m * Cin + g_rel -- no division in the loop
            // (the first version divided six times per lane and chunk: 0.48 ms for the first layer's weight gradient)
            unsigned v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ih = g_oh + g_dh[e], iw = g_ow + g_dw[e];
                const bool ok = ((unsigned)ih < (unsigned)a.Hin) & ((unsigned)iw < (unsigned)a.Win) & g_kv[e];   // m >= M: past num_records
                v[e] = __builtin_amdgcn_raw_buffer_load_b32(xr, ok ? (unsigned)(g_m * a.Cin + g_rel[e]) * 4u : OOB, 0, 0);
            }
            rb[0] = u32x4{v[0], v[1], v[2], v[3]};
            g_m += KP;
            g_ow += adv_w;
            { const bool c = g_ow >= a.Wl; g_ow -= c ? a.Wl : 0; g_oh += adv_h + (c ? 1 : 0); }
            { const bool c = g_oh >= a.Hl; g_oh -= c ? a.Hl : 0; }
        } else {
#pragma unroll
            for (int p = 0; p < BL; ++p)
                rb[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, rowtab[load_rel & 1][RPP * p + tid / (TN / 4)] + lane_col, 0, 0);
            ++load_rel;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int p = 0; p < AL; ++p) {
            int f = tid + NTH * p, row = f / (TM / 4), col = (f % (TM / 4)) * 4;
            *reinterpret_cast<u32x4*>(&As[buf][row * LDA + col]) = ra[p];
        }
        if constexpr (GATHER) {
            *reinterpret_cast<u32x4*>(&Bs[buf][(tid >> 3) * LDB + (tid & 7) * 4]) = rb[0];
        } else {
#pragma unroll
            for (int p = 0; p < BL; ++p) {
                int f = tid + NTH * p, row = f / (TN / 4), col = (f % (TN / 4)) * 4;
                *reinterpret_cast<u32x4*>(&Bs[buf][row * LDB + col]) = rb[p];
            }
        }
    };
    constexpr int KW = QUAD ? KP : KP / 4;  // pixels this wave consumes per chunk
    const int kbase = QUAD ? 0 : wave * KW;
    const int aoff = wm * WTM + (lane & 31), boff = wn * WTN + (lane & 31);
    // one k-pair (2 pixels): lanes 0-31 take pixel k, lanes 32-63 pixel k+1.  Fragments are double
    // buffered in registers: the LDS reads of pair i+1 are issued before the MFMAs of pair i.
    auto readfrag = [&](const float* __restrict__ Asm, const float* __restrict__ Bsm, int kk, float (&af)[MB], float (&bf)[NB]) {
        const int k = kbase + kk + (lane >> 5);
#pragma unroll
        for (int i = 0; i < MB; ++i) af[i] = Asm[k * LDA + aoff + i * 32];
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = Bsm[k * LDB + boff + j * 32];
    };
    auto mfma = [&](const float (&af)[MB], const float (&bf)[NB]) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    };

    constexpr int NP = KW / 2;   // k-pairs per chunk for this wave
    // U2 (plain operand forms): the chunk loop unrolled by two, so that the LDS buffer index is a compile-time constant
    // (wgrad_kernel<128,128,quad>: 13.86 -> 13.67 ms per step on MI355X; a second register set with the loads two chunks ahead,
    // the form the forward kernel uses, measured 13.85 with branches and 14.4 branch-free; the split forms are neutral)
    constexpr bool U2 = !GATHER;
    if constexpr (U2) {
        load();
        stage(0);
        __syncthreads();
        FV_WSTAMP(1);
        auto body = [&](int ch, auto odd) {
            constexpr bool ODD = decltype(odd)::value;
            const float* Ac = As[ODD ? 1 : 0]; const float* Bc = Bs[ODD ? 1 : 0];
            const bool more = ch + 1 < ch_end;
            if (more) load();
            publish_rows(ch - ch_begin + 2);
            float af0[MB], bf0[NB], af1[MB], bf1[NB];
            readfrag(Ac, Bc, 0, af0, bf0);
#pragma unroll
            for (int i = 0; i < NP; i += 2) {
                readfrag(Ac, Bc, 2 * (i + 1), af1, bf1);
                __builtin_amdgcn_sched_barrier(0);
                mfma(af0, bf0);
                if (i == NP / 2) { if (more) stage(ODD ? 0 : 1); }
                if (i + 2 < NP) readfrag(Ac, Bc, 2 * (i + 2), af0, bf0);
                __builtin_amdgcn_sched_barrier(0);
                mfma(af1, bf1);
            }
            __syncthreads();
        };
        for (int ch = ch_begin; ch < ch_end; ch += 2) {
            body(ch, std::false_type{});
            if (ch + 1 < ch_end) body(ch + 1, std::true_type{});
        }
    } else {
    // the gathered first layer: rolled chunk loop, the next chunk staged mid-chunk
    load();
    stage(0);
    __syncthreads();
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int cur = (ch - ch_begin) & 1;
        const bool more = ch + 1 < ch_end;
        if (more) load();
        float af0[MB], bf0[NB], af1[MB], bf1[NB];
        readfrag(As[cur], Bs[cur], 0, af0, bf0);
#pragma unroll
        for (int i = 0; i < NP; i += 2) {
            // sched_barrier: keep the reads of the next pair ahead of this pair's MFMAs (the scheduler
            // otherwise sinks them behind and waits lgkmcnt(0) in front of every MFMA group)
            readfrag(As[cur], Bs[cur], 2 * (i + 1), af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mfma(af0, bf0);
            if (i == NP / 2) { if (more) stage(cur ^ 1); }   // next tile lands in the other buffer mid-chunk
            if (i + 2 < NP) readfrag(As[cur], Bs[cur], 2 * (i + 2), af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            mfma(af1, bf1);
        }
        __syncthreads();
    }
    }

    FV_WSTAMP(2);
    const int half = lane >> 5, lc = lane & 31;
    const int Kw = GATHER ? 9 * a.Cin : 0;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int c = wn * WTN + j * 32 + lc;
                if (n < a.N) {
                    if constexpr (GATHER) {
                        if (c < Kw) atomicAdd(a.dw + (size_t)n * Kw + c, acc[i][j][r]);
                    } else {
                        atomicAdd(a.dw + ((size_t)n * a.Tw + a.taps.wslot[tp]) * a.Cin + c0 + c, acc[i][j][r]);
                    }
                }
            }
    FV_WSTAMP(3);
}

template <int TM, int TN, bool QUAD, bool GATHER>
int launch_w(fv_ctx* ctx, const FvWgradArgs& a) {
    const int ntap = GATHER ? 1 : a.taps.n;
    const int tiles = ((a.N + TM - 1) / TM) * (GATHER ? 1 : a.Cin / TN) * ntap;
    const int total_chunks = (a.M + KP - 1) / KP;
    // How many K-splits.  A workgroup pays a fixed cost (prologue + 64 float atomics per lane, about four
    // chunks' worth of time, and the atomics are L2 traffic), so prefer few, long splits whose workgroups
    // fill whole rounds of the resident slots (2 per CU); at least 8 chunks per workgroup, 16 for 1x1
    // kernels where all the parallelism comes from the split.
    //  * tiles < 64: splits come in groups of 8, one per XCD (see the kernel): rounds of 64 slots
    //  * tiles >= 64: a split already fills an XCD; plain order, rounds of all 512 slots
    // (round 5 re-measured the 1x1 bound: 12 or 8 chunks per workgroup fill all 512 slots and move the 52x52 / 26x26 / 13x13 launches
    // by +3 / -4 / +5 %: the atomics of twice as many workgroups cost what the idle slots cost)
    const int min_chunks = ntap == 1 ? 16 : 8;
    const int pinned = tiles < 64 ? 1 : 0;
    const int unit = pinned ? 8 : 1, slots = pinned ? 64 : 512;
    int qmax = total_chunks / (unit * min_chunks);
    qmax = qmax < 1 ? 1 : (qmax > 64 ? 64 : qmax);
    int best_q = 1;
    double best = -1.0;
    for (int q = 1; q <= qmax; ++q) {
        const double r = (double)q * tiles / slots, rounds = r <= 1.0 ? 1.0 : (double)(long long)(r + 0.999999);
        const double ch = (double)total_chunks / (unit * q);
        const double score = (r / rounds) * ch / (ch + 4.0);
        if (score > best * 1.005) { best = score; best_q = q; }
    }
    int nsplit = unit * best_q;
    if (nsplit > total_chunks) nsplit = total_chunks;   // tiny problems: empty splits return at once
    static const char* name = QUAD ? "wgrad_kernel<128,128,quad>"
                              : GATHER ? (TM == 64 ? "wgrad_kernel<64,32,gather>" : "wgrad_kernel<32,32,gather>")
                              : TM == 64 ? (TN == 64 ? "wgrad_kernel<64,64>" : "wgrad_kernel<64,32>")
                                         : (TN == 64 ? "wgrad_kernel<32,64>" : "wgrad_kernel<32,32>");
    FvProfScope ps(ctx, name, "M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " C" + std::to_string(a.Cin) + " t" + std::to_string(a.taps.n),
                   a.alg_flops, 4.0 * ((double)a.B * a.Hin * a.Win * a.Cin + (double)a.M * a.N + (double)a.N * a.Tw * a.Cin));
    const int nsplit8 = pinned ? (nsplit + 7) / 8 * 8 : nsplit;   // padded splits return at once (no chunks)
    if constexpr (QUAD) {
        if (ctx->conv_waves8)
            hipLaunchKernelGGL((wgrad_kernel<TM, TN, QUAD, GATHER, 8>), dim3(tiles * nsplit8), dim3(512), 0, ctx->stream, a, nsplit, ntap, tiles, pinned);
        else
            hipLaunchKernelGGL((wgrad_kernel<TM, TN, QUAD, GATHER>), dim3(tiles * nsplit8), dim3(256), 0, ctx->stream, a, nsplit, ntap, tiles, pinned);
    } else
        hipLaunchKernelGGL((wgrad_kernel<TM, TN, QUAD, GATHER>), dim3(tiles * nsplit8), dim3(256), 0, ctx->stream, a, nsplit, ntap, tiles, pinned);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

}  // namespace

#ifdef FV_CONV_STAMPS
extern "C" int fv_debug_wgrad_stamps(fv_ctx* ctx, unsigned long long* out, int nwg) {
    if (!ctx || !out || nwg < 1 || nwg > WSTAMP_WGS) return FV_ERR_INVALID;
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FV_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), (size_t)nwg * 5 * sizeof(unsigned long long)));
    return FV_OK;
}
#endif

int fv_wgrad_launch(fv_ctx* ctx, const FvWgradArgs& a) {
    FV_REQUIRE(ctx, a.x && a.dy && a.dw, "wgrad: NULL tensor");
    FV_REQUIRE(ctx, a.M > 0 && a.N > 0 && a.Ndy >= a.N && a.Ndy % 4 == 0, "wgrad: bad sizes (Ndy must be a multiple of 4)");
    FV_REQUIRE(ctx, (long long)a.B * a.Hin * a.Win * a.Cin < (1ll << 29) && (long long)a.M * a.Ndy < (1ll << 29),
               "wgrad: tensor exceeds 2^29 elements (2 GiB buffer descriptor)");
    if (a.Cin % 32 != 0) {
        FV_REQUIRE(ctx, 9 * a.Cin <= 32 && a.is == 1 && a.Hl == a.Hin && a.Wl == a.Win && a.taps.n == 9,
                   "wgrad: Cin=%d only supported as 3x3 stride-1 pad-1 with 9*Cin<=32", a.Cin);
        if (ctx->wgrad_fused_taps && fv_wgrad0_ok(a)) return fv_wgrad0_launch(ctx, a);
        if (a.N > 32) return launch_w<64, 32, false, true>(ctx, a);
        return launch_w<32, 32, false, true>(ctx, a);
    }
    FV_REQUIRE(ctx, a.taps.n >= 1 && a.taps.n <= 9, "wgrad: bad tap count");
    if (ctx->wgrad_fused_taps && fv_wgrad9_ok(a)) return fv_wgrad9_launch(ctx, a);
    if (ctx->wgrad_fused_taps && fv_wgrad1_ok(a)) return fv_wgrad1_launch(ctx, a);
    if (a.N >= 128 && a.Cin % 128 == 0) return launch_w<128, 128, true, false>(ctx, a);
    const bool n64 = a.N > 32, c64 = a.Cin % 64 == 0;
    if (n64 && c64) return launch_w<64, 64, false, false>(ctx, a);
    if (n64) return launch_w<64, 32, false, false>(ctx, a);
    if (c64) return launch_w<32, 64, false, false>(ctx, a);
    return launch_w<32, 32, false, false>(ctx, a);
}
