// Small-M inference: the split-K finish of a 3x3 layer FUSED with the 1x1 layer that follows it (one residual-block seam of the
// Darknet-53 base, reference yolov3_detect.py:226-267 as wired by face_detection.py:408-593; `self.model.predict`, fd.py:899).
//
// At batch 1 the per-layer path runs  conv(l) -> K-split slabs | splitk_finish (BN, LeakyReLU, + skip) | conv 1x1 (l+1)  as three
// dependent launches; the 1x1 layer has ~0.18 GFLOP (1 us of matrix time) and takes 13.5 us, the finish 6 us, each launch >= 3.7 us
// however little it does.  Here a workgroup owns 32 output pixels: it sums the slabs of ALL channels of those pixels in slab order
// (v = 0; v += slab_0; ...: the order of splitk_finish4_kernel), applies layer l's epilogue, stores the activation (the next block
// needs it as its skip tensor) and keeps it in LDS as the A operand of the 1x1 layer, which four of its waves then multiply on the
// matrix cores -- the same k-ordered fmaf chain per output element as conv_kernel<32,4,1> (within every eight channels lane-half h
// feeds k = 4h + j to MFMA j), so the result is bit-identical to the three-launch form.
#include <cstdlib>
#include "common.h"
#include "elementwise.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RBMAX = 32;       // rows of one MFMA block; a workgroup owns RB <= 32 pixels (the other rows of the block are zeros)
constexpr int NT2 = 128;        // output channels of the 1x1 layer per workgroup (4 waves x 32)
constexpr int NTH = 512;

struct FinishConvArgs {
    const float* slabs; int ks; long long sstride;       // [ks][M][C1], floats between slabs
    const float *scale1, *shift1, *skip1; float* out1;    // layer l: epilogue vectors [C1], residual addend [M][C1] or NULL, activation [M][C1]
    const float* w2; const float *scale2, *shift2; float* out2;     // layer l+1: kernel [C2][C1], epilogue vectors [C2], output [M][C2]
    int M, C2; float leaky;
};

template <int C1, int RB>
__global__ __launch_bounds__(NTH) void finish_conv1x1_kernel(const FinishConvArgs a) {
    constexpr int LDA = C1 + 4;                  // row pitch = 4 (mod 64) floats: the 16 lanes of a ds_read_b128 group hit 16 distinct 4-bank slots
    __shared__ __attribute__((aligned(16))) float As[RBMAX * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * RB, n0 = blockIdx.y * NT2;
    constexpr int C4 = C1 / 4;
    // ---- layer l: sum of the slabs in slab order, scale, shift, LeakyReLU, + skip (splitk_finish4_kernel's arithmetic)
    const bool writer = blockIdx.y == 0;
    for (int f = tid; f < RBMAX * C4; f += NTH) {
        const int row = f / C4, c = (f % C4) * 4;
        const int m = m0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < RB && m < a.M) {
            const float4* src = reinterpret_cast<const float4*>(a.slabs + (size_t)m * C1 + c);
            const size_t st4 = (size_t)(a.sstride >> 2);
            float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.skip1) sk = *reinterpret_cast<const float4*>(a.skip1 + (size_t)m * C1 + c);
            int k = 0;
            for (; k + 16 <= a.ks; k += 16) {
                float4 t[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) t[j] = src[(size_t)(k + j) * st4];
#pragma unroll
                for (int j = 0; j < 16; ++j) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
            }
            {
                float4 t[16];
                const int rem = a.ks - k;
#pragma unroll
                for (int j = 0; j < 16; ++j) t[j] = src[(size_t)(k + (j < rem ? j : (rem > 0 ? rem - 1 : -k))) * st4];
#pragma unroll
                for (int j = 0; j < 16; ++j) if (j < rem) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
            }
            { const float4 s = *reinterpret_cast<const float4*>(a.scale1 + c); v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
            { const float4 s = *reinterpret_cast<const float4*>(a.shift1 + c); v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
            v.x = v.x > 0.f ? v.x : v.x * a.leaky; v.y = v.y > 0.f ? v.y : v.y * a.leaky;
            v.z = v.z > 0.f ? v.z : v.z * a.leaky; v.w = v.w > 0.f ? v.w : v.w * a.leaky;
            if (a.skip1) { v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w; }
            if (writer) *reinterpret_cast<float4*>(a.out1 + (size_t)m * C1 + c) = v;
        }
        *reinterpret_cast<float4*>(&As[row * LDA + c]) = v;            // rows beyond M: zeros
    }
    __syncthreads();
    if (wave >= NT2 / 32) return;
    // ---- layer l+1 (1x1): out2[m][n] = sum_c a[m][c] * w2[n][c]; wave w owns columns n0 + 32 w .. + 31, B fragments straight from L2
    const int nl = wave * 32 + (lane & 31), n = n0 + nl;
    const int half = lane >> 5;
    const float* wrow = a.w2 + (size_t)(n < a.C2 ? n : 0) * C1 + half * 4;
    const float* arow = &As[(lane & 31) * LDA + half * 4];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NCH = C1 / 8;           // chunks of eight channels
    constexpr int PF = 8;                 // B fragments in flight
    float4 bq[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) bq[j] = *reinterpret_cast<const float4*>(wrow + j * 8);
#pragma unroll 1
    for (int c0 = 0; c0 < NCH; c0 += PF) {
        float4 bcur[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) bcur[j] = bq[j];
        if (c0 + PF < NCH) {
#pragma unroll
            for (int j = 0; j < PF; ++j) bq[j] = *reinterpret_cast<const float4*>(wrow + (c0 + PF + j) * 8);
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const float4 af = *reinterpret_cast<const float4*>(arow + (c0 + j) * 8);
            const float4 bf = bcur[j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
    }
    if (n >= a.C2) return;
    const float sc = a.scale2[n], sh = a.shift2[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m0 + row;
        if (row < RB && m < a.M) {
            float v = acc[r];
            v = v * sc + sh;
            v = v > 0.f ? v : v * a.leaky;
            a.out2[(size_t)m * a.C2 + n] = v;
        }
    }
}

}  // namespace

bool fv_ew_finish_conv1x1_ok(int C1, int C2, long long rows) {
    return (C1 == 256 || C1 == 512) && C2 * 2 == C1 && rows <= 4096;
}

static int g_rb = 0;   // FV_FINISH1X1_RB: pixels per workgroup (A/B knob); 0 = default
int fv_ew_finish_conv1x1(fv_ctx* ctx, const float* slabs, int ks, long long sstride, const float* scale1, const float* shift1,
                         const float* skip1, float* out1, const float* w2, const float* scale2, const float* shift2, float* out2,
                         int M, int C1, int C2, float leaky) {
    FV_REQUIRE(ctx, fv_ew_finish_conv1x1_ok(C1, C2, M) && ks >= 1 && (sstride & 3) == 0, "finish_conv1x1: unsupported shape");
    FV_REQUIRE(ctx, slabs && scale1 && shift1 && out1 && w2 && scale2 && shift2 && out2, "finish_conv1x1: NULL buffer");
    FvProfScope ps(ctx, "finish_conv1x1_kernel", 2.0 * M * C1 * (double)C2, 4.0 * ((double)M * C1 * (ks + 1 + (skip1 ? 1 : 0)) + (double)C1 * C2 + (double)M * C2));
    FinishConvArgs a{slabs, ks, sstride, scale1, shift1, skip1, out1, w2, scale2, shift2, out2, M, C2, leaky};
    if (g_rb == 0) { const char* e = getenv("FV_FINISH1X1_RB"); g_rb = e ? atoi(e) : 8; if (g_rb != 8 && g_rb != 16 && g_rb != 32) g_rb = 8; }     // measured at 416 x 416: 8 px 1.266 ms/img, 16 px 1.273, 32 px 1.378
    const int RB = g_rb;
    const dim3 grid((M + RB - 1) / RB, (C2 + NT2 - 1) / NT2);
#define FV_LAUNCH_FC(C, R) hipLaunchKernelGGL((finish_conv1x1_kernel<C, R>), grid, dim3(NTH), 0, ctx->stream, a)
    if (C1 == 256) { if (RB == 32) FV_LAUNCH_FC(256, 32); else if (RB == 16) FV_LAUNCH_FC(256, 16); else FV_LAUNCH_FC(256, 8); }
    else { if (RB == 32) FV_LAUNCH_FC(512, 32); else if (RB == 16) FV_LAUNCH_FC(512, 16); else FV_LAUNCH_FC(512, 8); }
#undef FV_LAUNCH_FC
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
