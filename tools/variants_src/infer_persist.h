// Internal interface of the one-launch small-M inference forward (infer_persist.hip), used by net.hip.
#pragma once
#include "common.h"

enum { FV_PERSIST_ERR_BARRIER = 1, FV_PERSIST_ERR_TILE = 2 };

// One conv layer of the persistent forward.  Offsets are in floats from the base they name, so the table depends on
// (batch, image_size, grid) only and is uploaded once per context.
struct FvPersistPhase {
    long long x_off;        // input activation, from the workspace
    long long w_off;        // kernel [cout][k][k][cin], from params
    long long out_off;      // output, from the workspace (out_sel 0) or from the external output pointer (1)
    long long skip_off;     // residual addend (layout of out), from the workspace; -1: none
    long long scale_off;    // per-channel scale, from the workspace; -1: none
    long long shift_off;    // per-channel shift, from the workspace (shift_sel 0) or params (1: the head's bias); -1: none
    int out_sel, shift_sel;
    int B, H, W, Cin, Cout, ksize, stride, Ho, Wo;
    int M;                  // B * Ho * Wo
    int nt, tiles;          // 128-wide column tiles, 128 x 128 tiles in all
    int ksplit, per;        // K slices per tile and K steps (of 32) per slice; (ksplit - 1) * per < ksteps <= ksplit * per
    int items;              // tiles * ksplit
    int cnt_off;            // first tile-arrival counter of this phase (index into the sync block's tile counters)
    int leaky_on;
    float leaky;
};

struct FvPersistArgs {
    const FvPersistPhase* table;    // device
    int nphase;
    float* ws;                      // workspace base
    const float* params;
    float* y_ext;                   // external output (head)
    long long slab_off;             // K-split partial slabs [item][128][128], from the workspace
    unsigned* sync;                 // sync block (zeroed before every launch)
    unsigned* err_host;             // pinned host copy of the error word (may be NULL)
    unsigned long long* trace;      // optional [3 * nphase + 1] wall-clock stamps of workgroup 0 (100 MHz); NULL: off
    double alg_flops;
    unsigned spin_limit;            // polls before a wait gives up (0: the default, ~1 s)
    int stall_wg;                   // test hook (FV_PERSIST_TEST_STALL): this workgroup leaves at the third barrier without arriving; -1: none
};

int fv_persist_sync_words(int total_tiles);
int fv_persist_max_grid(fv_ctx* ctx, int* blocks_per_cu, int* cus);
int fv_persist_launch(fv_ctx* ctx, const FvPersistArgs& a, int grid);
