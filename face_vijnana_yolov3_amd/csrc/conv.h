// Internal launch interface of the fp32-MFMA implicit-GEMM convolution kernels (conv_mfma.hip,
// wgrad_mfma.hip).  NHWC activations, OHWI-style weights ([n][tap][c], c contiguous).
#pragma once
#include "common.h"

// One "tap" = one (dh, dw) input offset and the slot of its weight slab.
struct FvTaps {
    int n;
    int dh[9];
    int dw[9];
    int wslot[9];
};

enum { FV_EPI_AFFINE = 1, FV_EPI_LEAKY = 2, FV_EPI_ADD = 4, FV_EPI_STATS = 8, FV_EPI_BNRED = 16 };

// Gather-convolution:  out[b, oh*os+oph, ow*os+opw, n] = sum_{t,c} x[b, oh*is+dh[t], ow*is+dw[t], c] * w[n][wslot[t]][c]
// over the output lattice (B, Hl, Wl); out-of-range input pixels read as zero.  This one form
// covers forward (stride 1/2, 3x3/1x1), data-gradient (stride 1: mirrored taps; stride 2: four
// parity classes, one per blockIdx.z) and the small-Cin first layer (gather mode, K = taps*Cin <= 32).
struct FvConvArgs {
    const float* x;
    const float* w;
    float* out;
    const float* addend;  // FV_EPI_ADD: same layout as out
    const float* scale;   // FV_EPI_AFFINE: per-n (may be NULL = 1)
    const float* shift;   // FV_EPI_AFFINE: per-n (may be NULL = 0)
    float* psum;          // FV_EPI_STATS: [mtiles][Nout] per-tile column sums of the raw result
    float* psq;           //                and of its squares
    double* stat_slots;   // FV_EPI_STATS, alternative to psum/psq: [stat_nslot][2][Nout] accumulators (sum, sum of
    int stat_nslot;       //   squares); tile mt ADDS its column sums to slot mt % stat_nslot (fp64 atomics)
    // FV_EPI_BNRED: the result is the gradient w.r.t. the output of a BN+LeakyReLU layer whose pre-BN tensor is bn_z
    // (same layout as out); the epilogue adds that layer's d-beta / d-gamma column sums to bn_slots [bn_nslot][2][Nout]
    const float *bn_z, *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
    double* bn_slots;
    int bn_nslot;
    float bn_leaky;
    int B, Hin, Win, Cin;
    int Hl, Wl;
    int Hout, Wout, Nout;
    int is, os;
    int Tw;       // taps per output channel in w
    int M;        // B*Hl*Wl
    int epi;
    float leaky;
    int nclass;   // 1, or 4 for stride-2 data-gradient
    double alg_flops;  // algorithmic 2*MAC of this launch (profiling only)
    int tail_f;        // >1: tail split active (set by the launcher): tiles >= tail_full are cut into tail_f K-slices
    int tail_full;
    float* tail_slab;  // [tail tiles * tail_f][128][BN] raw partial tiles
    int bm64;          // 1: 64-row tiles (fv_conv_bm64: small-M inference launches with 128-wide tiles)
    int small;         // 1..3: conv_small_kernel configuration (small-M inference launch, K split inside the workgroup; fv_conv_small_plan)
    int narrow;        // 1: 128x32 tiles whatever Nout (small-M 1x1 layers of the inference path, fv_conv_narrow)
    int ksplit;        // >1: blockIdx.y owns a slice of the K steps and stores its raw partial to out + y*split_stride
    long long split_stride;
    int oph[4], opw[4];
    FvTaps taps[4];
};

// Number of M tiles (rows of psum/psq) the conv launch will use for this problem.
int fv_conv_mtiles(int M, int Nout);
// K-split factor the small-M inference path uses for a problem (1 = no split).
int fv_conv_choose_ksplit(int M, int Nout, int ksteps, bool allow_bm64 = true);
// Small-M inference: 64-row instead of 128-row tiles for a launch with 128-wide tiles when that leaves fewer padded rows.
bool fv_conv_bm64(int M, int Nout, int ksteps);
// Small-M inference: a 1x1 layer (8..16 K steps) whose 128-wide tiling gives fewer than 64 tiles runs on 128x32 tiles WITHOUT a K
// split -- one launch instead of conv + split-K finish (a kernel costs >= 3.7 us on MI355X however little it does).
bool fv_conv_narrow(int M, int Nout, int ksteps);
// Tail split plan for a launch (M rows, Nout channels, ksteps K steps): slices per tail tile (1 = off),
// number of whole tiles, and the floats of slab scratch it needs.
void fv_conv_tail_plan(int M, int Nout, int ksteps, int* tail_f, int* tail_full, long long* slab_floats);
int fv_conv_launch(fv_ctx* ctx, const FvConvArgs& a);
// conv9_mfma.hip: training forward of the 32 -> 64 channel 3x3 layers from an LDS halo tile with resident weights (bit-identical z)
bool fv_conv9_fwd_ok(const FvConvArgs& a);
int fv_conv9_fwd_launch(fv_ctx* ctx, const FvConvArgs& a);
// dgrad9s2_mfma.hip: data-gradient (+ fused BN-backward reduction) of the stride-2 32 -> 64 channel 3x3 layer from a dy halo tile (bit-identical dx)
bool fv_dgrad9s2_ok(const FvConvArgs& a);
int fv_dgrad9s2_launch(fv_ctx* ctx, const FvConvArgs& a);
// conv1x1_mfma.hip: 1x1 stride-1 launches with more tiles than resident workgroup slots as a persistent GEMM (bit-identical)
bool fv_conv1x1_persist_ok(const FvConvArgs& a);
int fv_conv1x1_persist_launch(fv_ctx* ctx, const FvConvArgs& a);
// conv_small_mfma.hip: small-M inference launches (batch 1) with the K split inside the workgroup; plan: 0 = not taken, 1..3 = configuration
int fv_conv_small_plan(int M, int Nout, int Cin, int ntaps);
int fv_conv_small_launch(fv_ctx* ctx, const FvConvArgs& a);
// conv0_direct.hip: the 3 -> 32 channel first layer as a direct vector-FMA convolution (bit-identical to the gather kernel)
bool fv_conv0_direct_ok(const FvConvArgs& a);
int fv_conv0_direct_launch(fv_ctx* ctx, const FvConvArgs& a);

// Weight gradient:  dw[n][wslot[t]][c] += sum_{b,oh,ow} dy[b,oh,ow,n] * x[b, oh*is+dh[t], ow*is+dw[t], c]
// (dy is dense over the lattice (B,Hl,Wl) with Ndy channels per pixel of which the first N are used).
// Accumulates with float atomics: the caller zeroes dw first.
struct FvWgradArgs {
    const float* x;
    const float* dy;
    float* dw;
    int B, Hin, Win, Cin;  // x dims
    int Hl, Wl, N;         // dy lattice and used channels
    int Ndy;               // channel stride of dy (>= N)
    int is;
    int Tw;                // taps per output channel in dw
    int M;                 // B*Hl*Wl
    double alg_flops;      // algorithmic 2*MAC of this launch (profiling only)
    FvTaps taps;
};
int fv_wgrad_launch(fv_ctx* ctx, const FvWgradArgs& a);
// wgrad9_mfma.hip: the nine taps of a 3x3 layer with 32 -> 64 channels in one workgroup (x halo + dy tile staged once)
bool fv_wgrad0_ok(const FvWgradArgs& a);    // wgrad0_mfma.hip: the first layer (3 -> 32 channels)
int fv_wgrad0_launch(fv_ctx* ctx, const FvWgradArgs& a);
bool fv_wgrad1_ok(const FvWgradArgs& a);    // wgrad1_mfma.hip: the 1x1 layer with 64 -> 32 channels, streaming
int fv_wgrad1_launch(fv_ctx* ctx, const FvWgradArgs& a);
bool fv_wgrad9_ok(const FvWgradArgs& a);
int fv_wgrad9_launch(fv_ctx* ctx, const FvWgradArgs& a);
