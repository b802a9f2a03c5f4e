// Dev tool: sustained fp32-MFMA rate of this chip (v_mfma_f32_32x32x2_f32), to calibrate the
// roofline denominator: (a) registers only, (b) with the conv kernel's LDS fragment reads.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool LDS>
__global__ __launch_bounds__(256, 2) void peak(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 128 * 36];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) sm[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4 af[2], bf[2];
    af[0] = af[1] = bf[0] = bf[1] = make_float4(1.f + lane, 0.5f, 0.25f, 2.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            if (LDS) {
                const int ko = kc * 8 + (lane >> 5) * 4;
                af[0] = *reinterpret_cast<const float4*>(&sm[(lane & 31) * 36 + ko]);
                af[1] = *reinterpret_cast<const float4*>(&sm[(32 + (lane & 31)) * 36 + ko]);
                bf[0] = *reinterpret_cast<const float4*>(&sm[(128 + (lane & 31)) * 36 + ko]);
                bf[1] = *reinterpret_cast<const float4*>(&sm[(160 + (lane & 31)) * 36 + ko]);
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
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool LDS>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(peak<LDS>, dim3(blocks), dim3(256), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak<LDS>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 /*waves*/ * iters * 64.0 /*mfma*/ * 2 * 32 * 32 * 2;
    printf("%-28s blocks %4d  %.1f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<false>("registers only, 1 block/CU", 256);
    run<false>("registers only, 2 blocks/CU", 512);
    run<true>("with LDS reads, 1 block/CU", 256);
    run<true>("with LDS reads, 2 blocks/CU", 512);
    return 0;
}
