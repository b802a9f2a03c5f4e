// Dev tool: which ingredient of the conv K-step costs MFMA rate?  Variants add, per 64 MFMAs:
//  B = one __syncthreads, W = 8 ds_write_b128, G = 8 global float4 loads (L2-resident), all = B+W+G
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void peak(float* out, const float4* __restrict__ src, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[2][256 * 36];
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    for (int i = tid; i < 2 * 256 * 36; i += 256) (&sm[0][0])[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4 ld[8];
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        if (MODE & 4) {
#pragma unroll
            for (int p = 0; p < 8; ++p) ld[p] = src[(size_t)((blockIdx.x * 8 + p) * 256 + tid + (it & 63) * 4096) & 0xFFFFF];
        }
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int ko = kc * 8 + (lane >> 5) * 4;
            float4 af[2], bf[2];
            af[0] = *reinterpret_cast<const float4*>(&sm[cur][(lane & 31) * 36 + ko]);
            af[1] = *reinterpret_cast<const float4*>(&sm[cur][(32 + (lane & 31)) * 36 + ko]);
            bf[0] = *reinterpret_cast<const float4*>(&sm[cur][(128 + (lane & 31)) * 36 + ko]);
            bf[1] = *reinterpret_cast<const float4*>(&sm[cur][(160 + (lane & 31)) * 36 + ko]);
            if (kc == 2 && (MODE & 2)) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    float4 v = (MODE & 4) ? ld[p] : make_float4(1.f, 2.f, 3.f, (float)it);
                    *reinterpret_cast<float4*>(&sm[cur ^ 1][((tid >> 3) + 32 * p) * 36 + (tid & 7) * 4]) = v;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float av = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
                        const float bv = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i * 2 + j], 0, 0, 0);
                    }
        }
        if (MODE & 1) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if ((MODE & 4) && !(MODE & 2)) for (int p = 0; p < 8; ++p) s += ld[p].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, float4* src) {
    const int blocks = 512, iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(peak<MODE>, dim3(blocks), dim3(256), 0, 0, out, src, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(peak<MODE>, dim3(blocks), dim3(256), 0, 0, out, src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 64.0 * 2 * 32 * 32 * 2;
    printf("%-44s %.1f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main() {
    float* out; (void)hipMalloc(&out, 512 * 256 * 4);
    float4* src; (void)hipMalloc(&src, (size_t)(1 << 20) * 16); (void)hipMemset(src, 0, (size_t)(1 << 20) * 16);
    run<0>("LDS reads + MFMA", out, src);
    run<1>("+ barrier per 64 MFMA", out, src);
    run<2>("+ 8 ds_write_b128 per 64 MFMA", out, src);
    run<3>("+ barrier + ds_write", out, src);
    run<4>("+ 8 global loads per 64 MFMA", out, src);
    run<7>("+ barrier + ds_write + global loads (K-step)", out, src);
    return 0;
}
