// Which workgroups co-reside on a CU of gfx950?  Kernel A: one persistent 1024-thread workgroup per CU holding NV VGPRs per wave and LA bytes of
// LDS for ~1 ms.  Kernel B (second stream): 256-thread workgroups with NB VGPRs and LB bytes of LDS that only stamp their start time.  If B's first
// workgroup starts before A's last one ends, the two co-resided.     hipcc --offload-arch=gfx950 -O2 tools/coreside_micro.hip -o /tmp/cm && /tmp/cm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int NV>
__global__ __launch_bounds__(1024) void k_hold(unsigned long long* t, int lds_bytes, long long ticks) {
    extern __shared__ char sm[];
    float r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += r[i];
    if (lds_bytes > 0) sm[threadIdx.x % lds_bytes] = (char)s;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64(); }
    if (s == 12345.678f) t[0] = 0;
}

template <int NV>
__global__ __launch_bounds__(256) void k_stamp(unsigned long long* t, int lds_bytes) {
    extern __shared__ char sm[];
    float r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = wall_clock64();
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += r[i];
    if (lds_bytes > 0) sm[threadIdx.x % lds_bytes] = (char)s;
    if (threadIdx.x == 0) t[blockIdx.x] = t0;
    if (s == 12345.678f) t[0] = 0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NVA, int NVB>
void trial(int lds_a, int lds_b, int prio_b) {
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipStream_t sa, sb; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, prio_b ? hi : lo));
    unsigned long long *ta, *tb; CK(hipMalloc(&ta, 16 * cus)); CK(hipMalloc(&tb, 8 * cus));
    CK(hipFuncSetAttribute((const void*)k_hold<NVA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_stamp<NVB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_hold<NVA>, dim3(cus), dim3(1024), lds_a, sa, ta, lds_a, 100000LL);      // 1 ms at 100 MHz
        hipLaunchKernelGGL(k_stamp<NVB>, dim3(cus), dim3(256), lds_b, sb, tb, lds_b);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> ha(2 * cus), hb(cus);
    CK(hipMemcpy(ha.data(), ta, 16 * cus, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), tb, 8 * cus, hipMemcpyDeviceToHost));
    unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
    for (int i = 0; i < cus; ++i) { a0 = ha[2 * i] < a0 ? ha[2 * i] : a0; a1 = ha[2 * i + 1] > a1 ? ha[2 * i + 1] : a1; b0 = hb[i] < b0 ? hb[i] : b0; b1 = hb[i] > b1 ? hb[i] : b1; }
    int inside = 0; for (int i = 0; i < cus; ++i) inside += hb[i] < a1 - 1000;
    hipFuncAttributes fa, fb; CK(hipFuncGetAttributes(&fa, (const void*)k_hold<NVA>)); CK(hipFuncGetAttributes(&fb, (const void*)k_stamp<NVB>));
    printf("A: %3d VGPRs x 16 waves, %6d B LDS | B: %3d VGPRs x 4 waves, %6d B LDS, prio %s | A spans %.0f us, B workgroups start at %.0f .. %.0f us, %d of %d before A's end -> %s\n",
           fa.numRegs, lds_a, fb.numRegs, lds_b, prio_b ? "high" : "same", (a1 - a0) / 100.0, (double)(long long)(b0 - a0) / 100.0, (double)(long long)(b1 - a0) / 100.0, inside, cus,
           inside > cus / 2 ? "CO-RESIDENT" : "serialised");
    CK(hipFree(ta)); CK(hipFree(tb)); CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
}

int main() {
    trial<116, 20>(101392, 53248, 1);      // the split GEMM's footprint (120 VGPRs x 16 waves, 101 KiB) next to the stream aggregate's (<= 32 VGPRs, 52 KiB)
    trial<116, 24>(101392, 53248, 1);
    trial<116, 27>(101392, 53248, 1);
    trial<116, 28>(101392, 53248, 1);
    trial<116, 29>(101392, 53248, 1);
    trial<116, 30>(101392, 53248, 1);
    trial<116, 31>(101392, 53248, 1);
    trial<116, 34>(101392, 53248, 1);
    trial<112, 31>(101392, 53248, 1);
    trial<108, 31>(101392, 53248, 1);
    trial<108, 36>(101392, 53248, 1);
    trial<100, 36>(101392, 53248, 1);
    trial<88, 36>(101392, 53248, 1);
    trial<88, 50>(65536, 0, 1);
    return 0;
}
