// Persistent form of the 1x1 stride-1 convolution (forward of the 23 1x1 layers of yolov3_detect.py:221-267 and their
// data-gradients): a plain GEMM out[M][N] = x[M][K] . w[N][K]^T on the tile machinery of conv_mfma.hip.
//
// Why a second kernel (tools/conv_phases.py, round 5): a 1x1 tile lives 4 .. 16 K steps.  In the one-tile-per-workgroup launch
// all 512 resident workgroups run IN PHASE -- prologue (operand latency, nothing to multiply), K loop, epilogue -- over the
// whole chip: the matrix pipes idle during every prologue and epilogue, the memory system idles during every K loop (the
// data-gradients with the fused BN-backward reduction move 192 KB per tile in their epilogue), and a round of tiles costs
// prologue + K loop + epilogue although the 3x3 layers show that a workgroup's epilogue hides behind its CU partner's K loop
// once the two are out of phase.  Here a workgroup walks several tiles:
//  * the operand rows of tile i+1 (first two K steps) are in flight while tile i's epilogue runs -- no prologue latency after
//    the first tile, no workgroup launch between tiles;
//  * the two workgroups of a CU fall out of phase after their first tile (one wins the matrix pipe) and stay so:
//    one multiplies while the other stores;
//  * addressing is linear (row m of the lattice = row m of x and of out): no divisions, no row-offset table.
// Same tile shape, same K order, same MFMA sequence per output element as conv_kernel<BN, ...>: bit-identical results
// (tests/test_ops_gpu.py).  Tiles are dealt round-robin (workgroup b takes b, b + G, ...; the XCD remap keeps the N tiles of one
// A panel on one L2 at the same time).
#include <string>
#include <type_traits>
#include "conv_tile.h"

namespace {

#ifndef PF_BNRED
#define PF_BNRED 0
#endif

// PF: K steps of the NEXT tile whose operand loads are issued before the epilogue (0 .. 2; the rest right after it).  BNRED: the epilogue
// with the fused BN-backward reduction (its z / addend rows and per-channel vectors need the registers a deep prefetch would hold).
template <int BN, int WAVES_M, int WAVES_N, int PF, bool BNRED>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 4) void conv1x1_persist_kernel(const FvConvArgs a, const int ntiles) {
    constexpr int NTH = 64 * WAVES_M * WAVES_N;
    constexpr int APT = BM * 8 / NTH;
    constexpr int RSTEP = NTH / 8;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MB = WTM / 32, NB = WTN / 32;
    constexpr int BL = BN * 8 / NTH;
    static_assert(NTH == 512 && MB >= 1 && NB >= 1 && BL >= 1 && APT >= 1, "bad tiling");

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDT];
    static_assert(2 * (BM + BN) * LDT >= BM * BN + 2 * (NTH / 64) * BN, "operand LDS must hold the output tile + the BN-backward reduction scratch");
    float (*As)[BM * LDT] = reinterpret_cast<float (*)[BM * LDT]>(smem);
    float (*Bs)[BN * LDT] = reinterpret_cast<float (*)[BN * LDT]>(smem + 2 * BM * LDT);
    __shared__ float red[2][WAVES_M][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int NT = (a.Nout + BN - 1) / BN;
    const int nk = a.Cin / BK;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((unsigned)a.M * a.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((unsigned)a.Nout * a.Cin * 4u), 0x00020000);
    const int col4 = (tid & 7) * 4;
    unsigned a_off[APT], b_row[BL];
    u32x4 ra[APT], rb[BL], ra2[APT], rb2[BL];

    auto set_tile = [&](int w, int& mt, int& nt) {
        const int tile = xcd_remap(w, ntiles);
        mt = tile / NT; nt = tile - mt * NT;
#pragma unroll
        for (int p = 0; p < APT; ++p) {
            const int m = mt * BM + (tid >> 3) + RSTEP * p;
            a_off[p] = m < a.M ? (unsigned)(m * a.Cin + col4) * 4u : OOB;
        }
#pragma unroll
        for (int p = 0; p < BL; ++p) {
            const int n = nt * BN + (tid >> 3) + RSTEP * p;
            b_row[p] = n < a.Nout ? (unsigned)(n * a.Cin + col4) * 4u : OOB;
        }
    };
    auto load = [&](int step) {
        const int c0b = step * BK * 4;
#pragma unroll
        for (int p = 0; p < APT; ++p) ra[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, a_off[p], c0b, 0);
#pragma unroll
        for (int p = 0; p < BL; ++p) rb[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_row[p], c0b, 0);
    };
    auto load2 = [&](int step) {
        const int c0b = step * BK * 4;
#pragma unroll
        for (int p = 0; p < APT; ++p) ra2[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, a_off[p], c0b, 0);
#pragma unroll
        for (int p = 0; p < BL; ++p) rb2[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_row[p], c0b, 0);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int p = 0; p < APT; ++p) *reinterpret_cast<u32x4*>(&As[buf][((tid >> 3) + RSTEP * p) * LDT + col4]) = ra[p];
#pragma unroll
        for (int p = 0; p < BL; ++p) *reinterpret_cast<u32x4*>(&Bs[buf][((tid >> 3) + RSTEP * p) * LDT + col4]) = rb[p];
    };
    auto stage2 = [&](int buf) {
#pragma unroll
        for (int p = 0; p < APT; ++p) *reinterpret_cast<u32x4*>(&As[buf][((tid >> 3) + RSTEP * p) * LDT + col4]) = ra2[p];
#pragma unroll
        for (int p = 0; p < BL; ++p) *reinterpret_cast<u32x4*>(&Bs[buf][((tid >> 3) + RSTEP * p) * LDT + col4]) = rb2[p];
    };
    const int arow = (wm * WTM + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int brow = (wn * WTN + (lane & 31)) * LDT + (lane >> 5) * 4;
    auto readfrag = [&](const float* __restrict__ Asm, const float* __restrict__ Bsm, int kc, float4 (&af)[MB], float4 (&bf)[NB]) {
#pragma unroll
        for (int i = 0; i < MB; ++i) af[i] = *reinterpret_cast<const float4*>(&Asm[arow + i * 32 * LDT + kc * 8]);
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bsm[brow + j * 32 * LDT + kc * 8]);
    };

    int w = blockIdx.x, mt, nt;
    set_tile(w, mt, nt);
    load(0);
    if (nk > 1) load2(1);

    for (;;) {
        f32x16 acc[MB][NB];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
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
        stage(0);                 // K step 0 of this tile: loaded before the previous tile's epilogue (or above)
        __syncthreads();
        auto body = [&](int s, auto odd) {     // the K step of conv_kernel: same chunk order, same MFMA sequence
            constexpr bool ODD = decltype(odd)::value;
            const float* Ac = As[ODD ? 1 : 0]; const float* Bc = Bs[ODD ? 1 : 0];
            if (s + 2 < nk) { if constexpr (ODD) load2(s + 2); else load(s + 2); }
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
            if (s + 1 < nk) { if constexpr (ODD) stage(0); else stage2(1); }
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk(af1, bf1);
            __syncthreads();
        };
        for (int s = 0; s < nk; s += 2) {
            body(s, std::false_type{});
            if (s + 1 < nk) body(s + 1, std::true_type{});
        }

        // the next tile's first operand rows go out now and land while this tile is on its way out
        const int mt_c = mt, nt_c = nt;
        const int wnext = w + (int)gridDim.x;
        const bool more = wnext < ntiles;
        if (more) {
            set_tile(wnext, mt, nt);
            if constexpr (PF >= 1) load(0);
            if constexpr (PF >= 2) { if (nk > 1) load2(1); }
        }

        // ------------------------------------------------------------------ epilogue of tile (mt_c, nt_c)
        const int m0 = mt_c * BM, n0 = nt_c * BN;
        // the piece coordinates below depend on the thread id alone: left to itself the compiler hoists all of them out of the tile loop
        // and keeps ~40 VGPRs live through the K loop (164 instead of 122: one workgroup per CU).  An opaque copy of the thread id
        // makes them this iteration's values again.
        int te = tid;
        asm volatile("" : "+v"(te));
        const int lane_e = te & 63, half_e = lane_e >> 5, lc_e = lane_e & 31, wave_e = te >> 6;
        const int wm_e = wave_e / WAVES_N, wn_e = wave_e % WAVES_N;
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
                if (half_e == 0) { red[0][wm_e][wn_e * WTN + j * 32 + lc_e] = s; red[1][wm_e][wn_e * WTN + j * 32 + lc_e] = q; }
            }
            __syncthreads();
            if (te < BN && n0 + te < a.Nout) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int ww = 0; ww < WAVES_M; ++ww) { s += red[0][ww][te]; q += red[1][ww][te]; }
                stat_store(a, mt_c, n0 + te, s, q);
            }
        }
        float* Cs = smem;
        constexpr int C4 = BN / 4;
        constexpr int NP = BM * C4 / NTH, GP = NP < 4 ? NP : 4;
        static_assert(NP % GP == 0, "epilogue grouping");
        constexpr bool bnred = BNRED;
        const bool addon = (a.epi & FV_EPI_ADD) != 0;
        __syncthreads();         // (the K loop's last barrier already separates the operand reads from these writes; kept for the `red` reads above)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wm_e * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half_e) * BN + wn_e * WTN + j * 32 + lc_e] = acc[i][j][r];
        __syncthreads();
        BnRedAcc br;
        br.init(a, n0 + (te % C4) * 4, bnred && n0 + (te % C4) * 4 < a.Nout);
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += GP) {
            int offn[GP]; bool okp[GP];
            float4 zq[GP], aq[GP];
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const int f = te + NTH * (p0 + q), row = f / C4, c4 = (f % C4) * 4;
                const int m = m0 + row, n = n0 + c4;
                okp[q] = m < a.M && n < a.Nout;
                offn[q] = okp[q] ? m * a.Nout + n : 0;
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
                const int f = te + NTH * (p0 + q), row = f / C4, c4 = (f % C4) * 4;
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
                    *reinterpret_cast<float4*>(a.out + offn[q]) = v;
                    if (bnred) br.add(v, zq[q], a.bn_leaky);
                }
            }
        }
        if (bnred) bnred_flush<BN, NTH>(a, br, smem + BM * BN, n0, mt_c, te);
        if (!more) break;
        w = wnext;
        if constexpr (PF < 1) load(0);
        if constexpr (PF < 2) { if (nk > 1) load2(1); }
        __syncthreads();         // the output tile (and the reduction scratch behind it) has been read: the operand buffers are free again
    }
}

template <int BN, int WM_, int WN_>
int launch_persist(fv_ctx* ctx, const FvConvArgs& a) {
    const bool bnred = (a.epi & FV_EPI_BNRED) != 0;
    const int MT = (a.M + BM - 1) / BM, NT = (a.Nout + BN - 1) / BN;
    const int ntiles = MT * NT;
    const int grid = ntiles < 512 ? ntiles : 512;        // two workgroups per CU; a multiple of 8 whenever a workgroup takes a second tile
    static const std::string name_s = "conv1x1_persist_kernel<" + std::to_string(BN) + ">";
    static const char* name = name_s.c_str();
    FvProfScope ps(ctx, name, "M" + std::to_string(a.M) + " N" + std::to_string(a.Nout) + " K" + std::to_string(a.Cin) + ((a.epi & FV_EPI_BNRED) ? " r" : ""),
                   a.alg_flops, 4.0 * ((double)a.M * a.Cin + (double)a.Nout * a.Cin + (double)a.M * a.Nout * ((a.epi & FV_EPI_ADD) ? 2 : 1)));
    if (bnred) hipLaunchKernelGGL((conv1x1_persist_kernel<BN, WM_, WN_, PF_BNRED, true>), dim3(grid), dim3(64 * WM_ * WN_), 0, ctx->stream, a, ntiles);
    else hipLaunchKernelGGL((conv1x1_persist_kernel<BN, WM_, WN_, 2, false>), dim3(grid), dim3(64 * WM_ * WN_), 0, ctx->stream, a, ntiles);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

}  // namespace

// A 1x1 stride-1 launch over the whole lattice with 16-byte output rows and more tiles than the 512 resident slots: the case in
// which a workgroup of the persistent form gets a second tile.  (Fewer tiles: the one-tile kernel with its K split / tail split.)
bool fv_conv1x1_persist_ok(const FvConvArgs& a) {
    if (a.nclass != 1 || a.taps[0].n != 1 || a.taps[0].dh[0] != 0 || a.taps[0].dw[0] != 0 || a.taps[0].wslot[0] != 0 || a.Tw != 1) return false;
    if (a.is != 1 || a.os != 1 || a.Hl != a.Hin || a.Wl != a.Win || a.Hout != a.Hl || a.Wout != a.Wl || a.oph[0] || a.opw[0]) return false;
    if (a.Cin % BK != 0 || (a.Nout & 3) || a.Nout <= 32 || a.ksplit > 1 || a.narrow) return false;
    const int bn = a.Nout > 64 ? 128 : 64;
    const long long tiles = (long long)((a.M + BM - 1) / BM) * ((a.Nout + bn - 1) / bn);
    return tiles > 512;
}

int fv_conv1x1_persist_launch(fv_ctx* ctx, const FvConvArgs& a) {
    if (a.Nout > 64) return launch_persist<128, 2, 4>(ctx, a);
    return launch_persist<64, 4, 2>(ctx, a);
}
