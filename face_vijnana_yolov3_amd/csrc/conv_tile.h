// Pieces shared by the tile kernels of the fp32-MFMA gather convolution (conv_mfma.hip) and its persistent 1x1 form
// (conv1x1_mfma.hip): tile constants, the XCD-aware workgroup remap, the fused BN-backward reduction of the epilogue and the
// statistics store.  Device code in an anonymous namespace: include from a .hip file only.
#pragma once
#include "conv.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BK = 32;


constexpr int LDT = BK + 4;  // padded LDS row (dwords)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, pos = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + pos;
}

// Fused BatchNorm-backward reduction (FV_EPI_BNRED): the tile just produced is the gradient g w.r.t. the
// OUTPUT of a BN+LeakyReLU layer; with that layer's pre-BN tensor z the epilogue also forms
// gy = g * leaky'(z*scale+shift) and accumulates the column sums of gy and gy * xhat (d-beta, d-gamma)
// into that layer's fp64 slots -- the separate reduction pass over (g, z) disappears.
struct BnRedAcc {
    float4 sc, sh, mu, is, db, dg;
    __device__ __forceinline__ void init(const FvConvArgs& a, int n, bool on) {
        db = make_float4(0.f, 0.f, 0.f, 0.f); dg = db;
        sc = sh = mu = is = db;
        if (on) {
            sc = *reinterpret_cast<const float4*>(a.bn_scale + n); sh = *reinterpret_cast<const float4*>(a.bn_shift + n);
            mu = *reinterpret_cast<const float4*>(a.bn_mean + n); is = *reinterpret_cast<const float4*>(a.bn_invstd + n);
        }
    }
    __device__ __forceinline__ void add(const float4& g, const float4& z, float leaky) {
        float gy;
        gy = (z.x * sc.x + sh.x) > 0.f ? g.x : g.x * leaky; db.x += gy; dg.x += gy * ((z.x - mu.x) * is.x);
        gy = (z.y * sc.y + sh.y) > 0.f ? g.y : g.y * leaky; db.y += gy; dg.y += gy * ((z.y - mu.y) * is.y);
        gy = (z.z * sc.z + sh.z) > 0.f ? g.z : g.z * leaky; db.z += gy; dg.z += gy * ((z.z - mu.z) * is.z);
        gy = (z.w * sc.w + sh.w) > 0.f ? g.w : g.w * leaky; db.w += gy; dg.w += gy * ((z.w - mu.w) * is.w);
    }
};
// reduce the per-thread sums over the row lanes through LDS (scratch: 2 * RL * BN floats) and add the tile's column sums to
// slot `row_id % nslot`.  With more than 256 threads the two row lanes that share a wave (lanes l and l ^ 32 hold the same
// four columns when BN = 128) are combined by a shuffle first, so that the scratch still fits behind the output tile.
template <int BN, int NTH>
__device__ __forceinline__ void bnred_flush(const FvConvArgs& a, const BnRedAcc& r, float* scratch, int n0, int row_id, int tid) {
    constexpr int C4 = BN / 4;
    constexpr bool PAIR = NTH > 256;          // pre-reduce the row lanes that share a wave: one scratch row per wave
    static_assert(!PAIR || (64 % C4 == 0), "the in-wave pre-reduction needs the column groups to tile a wave");
    constexpr int RL = PAIR ? NTH / 64 : NTH / C4;
    const int c4 = (tid % C4) * 4;       // tid: threadIdx.x (a persistent caller passes an opaque copy, so that nothing here is hoisted out of its tile loop)
    float4 db = r.db, dg = r.dg;
    int rl = tid / C4;
    bool writer = true;
    if constexpr (PAIR) {
#pragma unroll
        for (int o = C4; o < 64; o <<= 1) {   // lanes l and l ^ o hold the same four columns
            db.x += __shfl_xor(db.x, o); db.y += __shfl_xor(db.y, o); db.z += __shfl_xor(db.z, o); db.w += __shfl_xor(db.w, o);
            dg.x += __shfl_xor(dg.x, o); dg.y += __shfl_xor(dg.y, o); dg.z += __shfl_xor(dg.z, o); dg.w += __shfl_xor(dg.w, o);
        }
        writer = (tid & 63) < C4;
        rl = tid >> 6;
    }
    float (*red)[RL][BN] = reinterpret_cast<float (*)[RL][BN]>(scratch);
    __syncthreads();
    if (writer) {
        *reinterpret_cast<float4*>(&red[0][rl][c4]) = db;
        *reinterpret_cast<float4*>(&red[1][rl][c4]) = dg;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Nout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < RL; ++w) { s += red[0][w][tid]; q += red[1][w][tid]; }
        double* sl = a.bn_slots + (size_t)(row_id % a.bn_nslot) * 2 * a.Nout;
        unsafeAtomicAdd(sl + n0 + tid, (double)s);
        unsafeAtomicAdd(sl + a.Nout + n0 + tid, (double)q);
    }
}

// Column sum / sum of squares of one tile: either its own partial row (deterministic; reduced later by
// bn_finalize) or added to one of a few fp64 accumulator slots (fp32 partials are exact in fp64; only
// the order of the fp64 additions varies, far below fp32 resolution) which the consumer kernel sums
// itself -- that saves the finalize launch between the conv and the normalise pass.
__device__ __forceinline__ void stat_store(const FvConvArgs& a, int mt, int n, float s, float q) {
    if (a.stat_slots) {
        double* sl = a.stat_slots + (size_t)(mt % a.stat_nslot) * 2 * a.Nout;
        unsafeAtomicAdd(sl + n, (double)s);
        unsafeAtomicAdd(sl + a.Nout + n, (double)q);
    } else {
        a.psum[(size_t)mt * a.Nout + n] = s;
        a.psq[(size_t)mt * a.Nout + n] = q;
    }
}

}  // namespace
