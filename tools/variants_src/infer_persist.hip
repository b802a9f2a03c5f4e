// Batch-1 (small-M) inference forward as ONE cooperative launch: layers 9 .. 51 of the Darknet-53 base and the head
// (reference face_detection.py:899 `self.model.predict(image)`, called one image at a time by detect()'s callers,
// face_identification.py:849, 1061).
//
// The per-layer path (net.hip) spends a batch-1 forward in ~85 dependent launches: at 52x52 .. 13x13 a layer has 8 - 85 tiles
// of 128x128, so its K loop is cut into slices, a second launch sums the slices' partial slabs, and every launch pays its own
// ramp, operand-latency prologue and drain.  Here 512 co-resident workgroups (hipLaunchCooperativeKernel: the runtime
// checks the grid against the occupancy) walk a per-layer work list:
//   phase A  item = (tile, K slice): the same 128 x 128 x 32 fp32-MFMA tile loop as conv_kernel<128,2,4> (conv_mfma.hip; same
//            operand staging, same k-ordered fmaf chain per output element).  An unsplit tile applies BN / LeakyReLU / skip and
//            stores; a slice stores its raw partial tile to a slab and ARRIVES on its tile's counter;
//   phase B  the workgroups that computed the slices of a tile wait for that counter (bounded) and each sums a band of the tile's
//            rows over all slabs in slice order -- the summation order of splitk_finish4_kernel -- and applies the epilogue;
//   barrier  an XCD-hierarchical grid barrier (per-group arrive counters -> top counter -> per-group generation words), bounded.
// Every wait is bounded: a workgroup that gives up raises the error word (device + pinned host copy) and leaves the kernel; the
// others see the word in their own polls and leave too, so the grid always drains.  Hand-offs follow the release / acquire
// recipe of the CDNA4 guide (every storing wave drains vmcnt, workgroup barrier, one agent-scope release, relaxed agent atomic;
// consumer: relaxed poll, one agent-scope acquire, workgroup barrier, plain loads).
#include <algorithm>
#include <cstring>
#include "conv.h"
#include "infer_persist.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;
constexpr int NTH = 512;
constexpr int WAVES_M = 2, WAVES_N = 4, WTM = BM / WAVES_M, WTN = BN / WAVES_N, MB = WTM / 32, NB = WTN / 32;
constexpr int APT = BM * 8 / NTH, BL = BN * 8 / NTH, RSTEP = NTH / 8;
static_assert(MB == 2 && NB == 1 && APT == 2 && BL == 2, "the tile loop below is written for 2 x 4 waves of 64 x 32");
constexpr unsigned SPIN_LIMIT = 1u << 21;       // polls of ~1 us each before a wait gives up
// Every byte handed from one workgroup to another inside the launch is stored write-through (sc1) and each storing wave drains its
// stores (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the arrive -- at that point the bytes are at the memory side,
// so the producer needs no release fence (nothing is dirty in its L2) [guide: "Valid forms", condition (2) + (3)].  Consumers: the
// slabs are read with sc1 loads (past the L1); activations are read with plain loads behind the ONE agent-scope acquire of the
// grid barrier.  FV_PERSIST_FENCES=1 builds the belt-and-braces form (release fence before every arrive, acquire after every wait).
#ifndef FV_PERSIST_FENCES
#define FV_PERSIST_FENCES 0
#endif

// sync block (unsigned words; every polled word on a 128-byte line of its own)
constexpr int SYNC_STRIDE = 32;
constexpr int SYNC_ARRIVE = 0;                      // [8] per-group arrive counters
constexpr int SYNC_TOP = 8 * SYNC_STRIDE;
constexpr int SYNC_GEN = 9 * SYNC_STRIDE;           // [8] per-group generation words
constexpr int SYNC_ERR = 17 * SYNC_STRIDE;
constexpr int SYNC_TILES = 18 * SYNC_STRIDE;        // per-phase tile arrival counters follow

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, pos = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + pos;
}
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lane 0 of a workgroup: poll *p until it reaches `target`.  false: gave up (limit, or another workgroup raised the error word)
__device__ bool spin_until_ge(const unsigned* p, unsigned target, unsigned* err, unsigned limit) {
    for (unsigned it = 0;; ++it) {
        if (ld_relaxed(p) >= target) return true;
        if (it >= limit) return false;
        if ((it & 31) == 31 && ld_relaxed(err) != 0) return false;
        __builtin_amdgcn_s_sleep(16);           // ~0.4 us between polls: a poll is a fabric request on a line other CUs are adding to
    }
}
__device__ __forceinline__ void raise_error(const FvPersistArgs& A, unsigned code) {
    atomicOr(A.sync + SYNC_ERR, code);
    if (A.err_host) __hip_atomic_store(A.err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Grid barrier number `epoch` (1, 2, ...) of this launch.  Groups = blockIdx.x & 7 (under the observed round-robin placement the
// workgroups of one XCD; correctness does not depend on it: a group is defined by the index, its size is gridDim.x / 8).
// Counters are monotonic within the launch (zeroed by a memset node before it).  Returns false when the wait was abandoned.
__device__ bool grid_barrier(const FvPersistArgs& A, unsigned epoch, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its own stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7, per_group = gridDim.x >> 3;
#if FV_PERSIST_FENCES
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned old = __hip_atomic_fetch_add(A.sync + SYNC_ARRIVE + g * SYNC_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch * per_group) {                // last of its group
            const unsigned o2 = __hip_atomic_fetch_add(A.sync + SYNC_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1 == epoch * 8u) {                    // last group: open the gate of every group
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __hip_atomic_store(A.sync + SYNC_GEN + k * SYNC_STRIDE, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const bool ok = spin_until_ge(A.sync + SYNC_GEN + g * SYNC_STRIDE, epoch, A.sync + SYNC_ERR, A.spin_limit ? A.spin_limit : SPIN_LIMIT);
        if (!ok) raise_error(A, FV_PERSIST_ERR_BARRIER);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_flag = ok ? 1 : 0;
    }
    __syncthreads();
    return *s_flag != 0;
}

// The phase table is read through the constant address space: uniform scalar loads (s_load), whatever the kernel stored before.
template <typename T>
__device__ __forceinline__ T ldc(const T* p) {
    typedef const T __attribute__((address_space(4))) * CP;
    return *(CP)(uintptr_t)p;
}
__device__ __forceinline__ FvPersistPhase load_phase(const FvPersistPhase* t) {
    FvPersistPhase P;
    P.x_off = ldc(&t->x_off); P.w_off = ldc(&t->w_off); P.out_off = ldc(&t->out_off); P.skip_off = ldc(&t->skip_off);
    P.scale_off = ldc(&t->scale_off); P.shift_off = ldc(&t->shift_off); P.out_sel = ldc(&t->out_sel); P.shift_sel = ldc(&t->shift_sel);
    P.B = ldc(&t->B); P.H = ldc(&t->H); P.W = ldc(&t->W); P.Cin = ldc(&t->Cin); P.Cout = ldc(&t->Cout); P.ksize = ldc(&t->ksize);
    P.stride = ldc(&t->stride); P.Ho = ldc(&t->Ho); P.Wo = ldc(&t->Wo); P.M = ldc(&t->M); P.nt = ldc(&t->nt); P.tiles = ldc(&t->tiles);
    P.ksplit = ldc(&t->ksplit); P.per = ldc(&t->per); P.items = ldc(&t->items); P.cnt_off = ldc(&t->cnt_off);
    P.leaky_on = ldc(&t->leaky_on); P.leaky = ldc(&t->leaky);
    return P;
}

// Hand-off traffic is written THROUGH the XCD's L2 (sc1): nothing is left dirty for the release fences to write back (the guide's
// "publish-large" row: 64 KB per workgroup, 3.0 us with write-through stores against 8.2 us with plain stores + release), and the
// slabs are read with sc1 loads (served by the L2, past this CU's L1).
constexpr int AUX_SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void st4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const float4& v) {
    u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, AUX_SC1);
}
__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX_SC1);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}

struct Bases {
    const float* x; const float* w; float* out; const float* skip; const float* scale; const float* shift; float* slab;
};
__device__ __forceinline__ Bases resolve(const FvPersistArgs& A, const FvPersistPhase& P) {
    Bases b;
    b.x = A.ws + P.x_off;
    b.w = A.params + P.w_off;
    b.out = (P.out_sel ? A.y_ext : A.ws) + P.out_off;
    b.skip = P.skip_off >= 0 ? A.ws + P.skip_off : nullptr;
    b.scale = P.scale_off >= 0 ? A.ws + P.scale_off : nullptr;     // (ws is written by earlier phases, never at these offsets)
    b.shift = P.shift_off >= 0 ? (P.shift_sel ? A.params : A.ws) + P.shift_off : nullptr;
    b.slab = A.ws + A.slab_off;
    return b;
}

__device__ __forceinline__ float4 epilogue4(float4 v, const FvPersistPhase& P, const Bases& b, int n, long long off) {
    if (b.scale) { const float4 s = *reinterpret_cast<const float4*>(b.scale + n); v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
    if (b.shift) { const float4 s = *reinterpret_cast<const float4*>(b.shift + n); v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
    if (P.leaky_on) {
        v.x = v.x > 0.f ? v.x : v.x * P.leaky; v.y = v.y > 0.f ? v.y : v.y * P.leaky;
        v.z = v.z > 0.f ? v.z : v.z * P.leaky; v.w = v.w > 0.f ? v.w : v.w * P.leaky;
    }
    if (b.skip) { const float4 s = *reinterpret_cast<const float4*>(b.skip + off + n); v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
    return v;
}

// Phase A item: K steps [slice * per, ...) of output tile `tile`.  The tile loop is conv_kernel<128, 2, 4, false>'s (conv_mfma.hip),
// specialised to the forward tap list of a 3x3 / 1x1 layer: same loads, same LDS image, same MFMA order.
__device__ __forceinline__ void conv_item(const FvPersistArgs& A, const FvPersistPhase& P, const Bases& b, int tile, int slice,
                                          float* smem, int* rowoff) {
    float (*As)[BM * LDT] = reinterpret_cast<float (*)[BM * LDT]>(smem);
    float (*Bs)[BN * LDT] = reinterpret_cast<float (*)[BN * LDT]>(smem + 2 * BM * LDT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int mt = tile / P.nt, nt = tile - mt * P.nt;
    const int m0 = mt * BM, n0 = nt * BN;
    const int HWl = P.Ho * P.Wo;
    __syncthreads();                       // the previous item / reduce pass of this workgroup is done with smem and rowoff
    if (tid < BM) {
        int m = m0 + tid, off = -1;
        if (m < P.M) {
            int bb = m / HWl, rem = m - bb * HWl, oh = rem / P.Wo, ow = rem - oh * P.Wo;
            off = ((bb * P.Ho + oh) * P.Wo + ow) * P.Cout;
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

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)b.x, 0, (int)((unsigned)P.B * P.H * P.W * P.Cin * 4u), 0x00020000);
    const int Tw = P.ksize * P.ksize;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)b.w, 0, (int)((unsigned)P.Cout * Tw * P.Cin * 4u), 0x00020000);
    const int col4 = (tid & 7) * 4;
    int a_pix[APT], a_oh[APT], a_ow[APT];
#pragma unroll
    for (int p = 0; p < APT; ++p) {
        int m = m0 + (tid >> 3) + RSTEP * p;
        if (m < P.M) {
            int bb = m / HWl, rem = m - bb * HWl, oh = rem / P.Wo, ow = rem - oh * P.Wo;
            a_pix[p] = bb * P.H; a_oh[p] = oh * P.stride; a_ow[p] = ow * P.stride;
        } else {
            a_pix[p] = 0; a_oh[p] = -(1 << 28); a_ow[p] = 0;
        }
    }
    unsigned b_row[BL];
#pragma unroll
    for (int p = 0; p < BL; ++p) {
        int n = n0 + (tid >> 3) + RSTEP * p;
        b_row[p] = n < P.Cout ? (unsigned)(n * Tw * P.Cin + col4) * 4u : OOB;
    }
    const int cpk = P.Cin / BK;
    const int nk = Tw * cpk;
    const int s_begin = slice * P.per;
    const int s_end = min(nk, s_begin + P.per);
    u32x4 ra[APT], rb[BL], ra2[APT], rb2[BL];
    unsigned a_off[APT];
    int t = s_begin / cpk, ci = s_begin - t * cpk;
    const int k3 = P.ksize == 3;
    auto set_tap = [&](int tp) {
        const int r3 = tp / 3;
        const int dh = k3 ? r3 - 1 : 0, dw = k3 ? tp - 3 * r3 - 1 : 0;
#pragma unroll
        for (int p = 0; p < APT; ++p) {
            int ih = a_oh[p] + dh, iw = a_ow[p] + dw;
            bool ok = (unsigned)ih < (unsigned)P.H && (unsigned)iw < (unsigned)P.W;
            a_off[p] = ok ? (unsigned)(((a_pix[p] + ih) * P.W + iw) * P.Cin + col4) * 4u : OOB;
        }
    };
    auto load = [&]() {
        const int c0b = ci * BK * 4;
        const int wofs = (t * P.Cin) * 4 + c0b;
#pragma unroll
        for (int p = 0; p < APT; ++p) ra[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, a_off[p], c0b, 0);
#pragma unroll
        for (int p = 0; p < BL; ++p) rb[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_row[p], wofs, 0);
    };
    auto load2 = [&]() {
        const int c0b = ci * BK * 4;
        const int wofs = (t * P.Cin) * 4 + c0b;
#pragma unroll
        for (int p = 0; p < APT; ++p) ra2[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, a_off[p], c0b, 0);
#pragma unroll
        for (int p = 0; p < BL; ++p) rb2[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_row[p], wofs, 0);
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
    auto advance = [&]() {
        if (++ci == cpk) { ci = 0; ++t; if (t < Tw) set_tap(t); }
    };
    const int arow = (wm * WTM + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int brow = (wn * WTN + (lane & 31)) * LDT + (lane >> 5) * 4;
    auto readfrag = [&](const float* __restrict__ Asm, const float* __restrict__ Bsm, int kc, float4 (&af)[MB], float4 (&bf)[NB]) {
#pragma unroll
        for (int i = 0; i < MB; ++i) af[i] = *reinterpret_cast<const float4*>(&Asm[arow + i * 32 * LDT + kc * 8]);
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bsm[brow + j * 32 * LDT + kc * 8]);
    };
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
    if (s_begin < s_end) {
        set_tap(t);
        load(); advance();
        if (s_begin + 1 < s_end) { load2(); advance(); }
        stage(0);
    }
    __syncthreads();
    auto body = [&](int s, auto odd) {
        constexpr bool ODD = decltype(odd)::value;
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

    // ---- epilogue: transpose the tile through the operand LDS, then 16-byte pieces
    const int half = lane >> 5, lc = lane & 31;
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
    constexpr int C4 = BN / 4;
    if (P.ksplit > 1) {
        // raw partial tile, tile-local layout, to this item's slab; then arrive on the tile's counter
        const __amdgpu_buffer_rsrc_t sr = rsrc_of(b.slab + (size_t)(slice * P.tiles + tile) * (BM * BN), BM * BN * 4u);
#pragma unroll
        for (int p = 0; p < BM * C4 / NTH; ++p) st4_sc1(sr, (unsigned)(tid + NTH * p) * 16u, reinterpret_cast<const float4*>(Cs)[tid + NTH * p]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
#if FV_PERSIST_FENCES
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __hip_atomic_fetch_add(A.sync + SYNC_TILES + (P.cnt_off + tile) * SYNC_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t orr = rsrc_of(b.out, (unsigned)P.M * P.Cout * 4u);
#pragma unroll
    for (int p = 0; p < BM * C4 / NTH; ++p) {
        const int f = tid + NTH * p, row = f / C4, c4 = (f % C4) * 4;
        const int off = rowoff[row], n = n0 + c4;
        if (off >= 0 && n < P.Cout) {
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * BN + c4]);
            v = epilogue4(v, P, b, n, off);
            st4_sc1(orr, (unsigned)(off + n) * 4u, v);
        }
    }
}

// Phase B: the workgroup that computed slice `slice` of `tile` sums rows [slice * 128 / ks, (slice + 1) * 128 / ks) of the tile over
// all ks slabs, in slice order (v = 0; v += slab_0; v += slab_1; ... -- the order of splitk_finish4_kernel), then scale, shift,
// LeakyReLU, skip, store.  Sixteen slab loads are in flight per thread.
__device__ __forceinline__ bool reduce_item(const FvPersistArgs& A, const FvPersistPhase& P, const Bases& b, int tile, int slice, int* s_flag) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        const bool ok = spin_until_ge(A.sync + SYNC_TILES + (P.cnt_off + tile) * SYNC_STRIDE, (unsigned)P.ksplit, A.sync + SYNC_ERR,
                                      A.spin_limit ? A.spin_limit : SPIN_LIMIT);
        if (!ok) raise_error(A, FV_PERSIST_ERR_TILE);
#if FV_PERSIST_FENCES
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        *s_flag = ok ? 1 : 0;
    }
    __syncthreads();
    if (*s_flag == 0) return false;
    const int ks = P.ksplit;
    const int r0 = slice * BM / ks, r1 = (slice + 1) * BM / ks;
    const int mt = tile / P.nt, nt = tile - mt * P.nt;
    const int m0 = mt * BM, n0 = nt * BN;
    const int HWl = P.Ho * P.Wo;
    constexpr int C4 = BN / 4;
    const __amdgpu_buffer_rsrc_t sr = rsrc_of(b.slab, (unsigned)P.items * (BM * BN * 4u));      // <= 1024 items: < 2^26 bytes
    const __amdgpu_buffer_rsrc_t orr = rsrc_of(b.out, (unsigned)P.M * P.Cout * 4u);
    const unsigned sstride = (unsigned)P.tiles * (BM * BN * 4u);                                 // bytes between the slices of a tile
    for (int f = tid; f < (r1 - r0) * C4; f += NTH) {
        const int row = r0 + f / C4, c4 = (f % C4) * 4;
        const int m = m0 + row, n = n0 + c4;
        if (m >= P.M || n >= P.Cout) continue;
        const int bb = m / HWl, rem = m - bb * HWl, oh = rem / P.Wo, ow = rem - oh * P.Wo;
        const long long off = (long long)((bb * P.Ho + oh) * P.Wo + ow) * P.Cout;
        const unsigned src = (unsigned)tile * (BM * BN * 4u) + (unsigned)(row * BN + c4) * 4u;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 16 <= ks; k += 16) {
            float4 tt[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) tt[j] = ld4_sc1(sr, src + (unsigned)(k + j) * sstride);
#pragma unroll
            for (int j = 0; j < 16; ++j) { v.x += tt[j].x; v.y += tt[j].y; v.z += tt[j].z; v.w += tt[j].w; }
        }
        {
            float4 tt[16];
            const int rem_k = ks - k;
#pragma unroll
            for (int j = 0; j < 16; ++j) tt[j] = ld4_sc1(sr, src + (unsigned)(k + (j < rem_k ? j : (rem_k > 0 ? rem_k - 1 : -k))) * sstride);
#pragma unroll
            for (int j = 0; j < 16; ++j) if (j < rem_k) { v.x += tt[j].x; v.y += tt[j].y; v.z += tt[j].z; v.w += tt[j].w; }
        }
        if ((P.Cout & 3) == 0) {
            v = epilogue4(v, P, b, n, off);
            st4_sc1(orr, (unsigned)(off + n) * 4u, v);
        } else {                                   // the head: 6 channels, rows not 16-byte aligned
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (n + q >= P.Cout) break;
                float u = e[q];
                if (b.scale) u *= b.scale[n + q];
                if (b.shift) u += b.shift[n + q];
                if (P.leaky_on) u = u > 0.f ? u : u * P.leaky;
                if (b.skip) u += b.skip[off + n + q];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u), orr, (unsigned)(off + n + q) * 4u, 0, AUX_SC1);
            }
        }
    }
    return true;
}

// (NTH, 2): two waves per SIMD = ONE workgroup per CU, the configuration that is run (net.hip persist_forward): 151 VGPRs, no scratch;
// with (NTH, 4) the 128-VGPR cap spilled 18 VGPRs (76 B/lane of scratch) into the tile loop
__global__ __launch_bounds__(NTH, 2) void infer_persist_kernel(const FvPersistArgs A) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDT];
    __shared__ int rowoff[BM];
    __shared__ int s_flag;
    const int G = gridDim.x;
    for (int ph = 0; ph < A.nphase; ++ph) {
        const FvPersistPhase P = load_phase(A.table + ph);
        const Bases b = resolve(A, P);
        if (A.trace && blockIdx.x == 0 && threadIdx.x == 0) A.trace[3 * ph] = wall_clock64();
        for (int it = blockIdx.x; it < P.items; it += G) {
            const int slice = it / P.tiles, tb = it - slice * P.tiles;
            conv_item(A, P, b, xcd_remap(tb, P.tiles), slice, smem, rowoff);
        }
        if (A.trace && blockIdx.x == 0 && threadIdx.x == 0) A.trace[3 * ph + 1] = wall_clock64();
        if (P.ksplit > 1) {
            for (int it = blockIdx.x; it < P.items; it += G) {
                const int slice = it / P.tiles, tb = it - slice * P.tiles;
                if (!reduce_item(A, P, b, xcd_remap(tb, P.tiles), slice, &s_flag)) return;
            }
        }
        if (A.trace && blockIdx.x == 0 && threadIdx.x == 0) A.trace[3 * ph + 2] = wall_clock64();
        if (ph == 2 && (int)blockIdx.x == A.stall_wg) return;      // test hook: a workgroup that is lost before a barrier
        if (ph + 1 < A.nphase && !grid_barrier(A, (unsigned)(ph + 1), &s_flag)) return;
    }
    if (A.trace && blockIdx.x == 0 && threadIdx.x == 0) A.trace[3 * A.nphase] = wall_clock64();
}

}  // namespace

int fv_persist_sync_words(int total_tiles) { return SYNC_TILES + total_tiles * SYNC_STRIDE; }

int fv_persist_max_grid(fv_ctx* ctx, int* blocks_per_cu, int* cus) {
    int nb = 0;
    FV_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, infer_persist_kernel, NTH, 0));
    hipDeviceProp_t prop;
    FV_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (blocks_per_cu) *blocks_per_cu = nb;
    if (cus) *cus = prop.multiProcessorCount;
    return FV_OK;
}

int fv_persist_launch(fv_ctx* ctx, const FvPersistArgs& a, int grid) {
    FV_REQUIRE(ctx, grid >= 8 && grid % 8 == 0, "persist: the grid must be a multiple of 8 workgroups");
    FvProfScope ps(ctx, "infer_persist_kernel", a.alg_flops, 0.0);
    FvPersistArgs args = a;
    if (ctx->persist_plain_launch) {
        // Same residency as the cooperative launch (the guide: plain, cooperative and graph launches place a grid alike); what is given up
        // is the runtime's own check of the grid against the occupancy -- done here instead -- and what is gained is that the kernel stays
        // on the stream's own hardware queue (a cooperative launch goes through the device's cooperative queue and is ordered against
        // every other queue of the process: 2.1 instead of 1.3 ms per forward once an RCCL communicator has existed in the process).
        int per_cu = 0, cus = 0;
        if (int rc = fv_persist_max_grid(ctx, &per_cu, &cus)) return rc;
        if (per_cu < 1 || grid > per_cu * cus) return fv_fail(ctx, FV_ERR_HIP, "persist: a grid of %d workgroups is not co-resident (%d per CU x %d CUs)", grid, per_cu, cus);
        hipLaunchKernelGGL(infer_persist_kernel, dim3(grid), dim3(NTH), 0, ctx->stream, args);
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    }
    void* kargs[] = {(void*)&args};
    FV_HIP(ctx, hipLaunchCooperativeKernel((const void*)infer_persist_kernel, dim3(grid), dim3(NTH), kargs, 0, ctx->stream));
    return FV_OK;
}
