// Standalone bench + accuracy check of the persistent split GEMM with NP = 3 (three bf16 pieces, six products) and NP = 2 (two fp16 pieces under
// per-set power-of-two scales, three products).   hipcc --offload-arch=gfx950 -O3 -I g-meta_amd/csrc tools/gemm_f16_bench.hip -o tools/_build/gf16
//   gf16 M K N NP [range_bits]     GS_SLEEP_US=2000: cool-chip mode.  range_bits r: row i of A is scaled by 2^-(i % (r+1)) (dynamic range inside one set)
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_split.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1146880, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 256;
    const int NP = argc > 4 ? atoi(argv[4]) : 2, range = argc > 5 ? atoi(argv[5]) : 0;
    printf("M=%d K=%d N=%d NP=%d range_bits=%d\n", M, K, N, NP, range);
    std::vector<float> A((size_t)M * K), W((size_t)K * N), bias(N);
    srand(1);
    auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
    for (auto& v : W) v = rnd() * 0.1f;
    for (auto& v : bias) v = rnd() * 1e-3f;
    for (size_t i = 0; i < A.size(); ++i) A[i] = ldexpf(rnd() * (1.f + (i % 7)), range ? -(int)((i / K) % (range + 1)) : 0);
    std::vector<int32_t> tiles;
    for (int r = 0; r < M; r += 128) { tiles.push_back(0); tiles.push_back(r); tiles.push_back(std::min(128, M - r)); }
    float *dA, *dW, *dC, *dBias; uint16_t* dBt; int32_t* dT; unsigned* dMax;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dBias, N * 4));
    CK(hipMalloc(&dBt, (size_t)3 * N * K * 2)); CK(hipMalloc(&dT, tiles.size() * 4)); CK(hipMalloc(&dMax, 64)); CK(hipMemset(dMax, 0, 64));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBias, bias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dT, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_amax, dim3(1024, 1), dim3(256), 0, 0, dA, 0, 0, (int64_t)M * K, dMax + 0, 0);
    hipLaunchKernelGGL(k_amax, dim3(64, 1), dim3(256), 0, 0, dW, 0, 0, (int64_t)K * N, dMax + 1, 0);
    hipLaunchKernelGGL(k_split_w, dim3((K + 31) / 32, (N + 31) / 32, 1), dim3(256), 0, 0, dW, 0, 0, K, N, 0, dBt, NP, gm_bound{dMax + 1, 0, nullptr, 1.f});
    SplitGemmK g{};
    g.A = dA; g.lda = K; g.Bt = dBt; g.bt_stride = 0; g.C = dC; g.ldc = N; g.K = K; g.N = N; g.bias = dBias; g.relu = 0;
    g.tiles = dT; g.n_tiles = (int)tiles.size() / 3; g.n_col_tiles = 1; g.nt_store = getenv("GS_NT") ? atoi(getenv("GS_NT")) : 1;
    g.a_bound = gm_bound{dMax, 0, nullptr, 1.f}; g.b_bound = gm_bound{dMax + 1, 0, nullptr, 1.f}; g.amax_out = getenv("GS_NOAMAX") ? nullptr : dMax + 2;
    const int cap = getenv("GS_GRID") ? atoi(getenv("GS_GRID")) : 256;
    auto launch = [&]() {
        if (N == 256 && NP == 2) hipLaunchKernelGGL((k_gemm_split_p<false, 2, 4, 2>), dim3(std::min(g.n_tiles, cap)), dim3(1024), 0, 0, g);
        else if (N == 256) hipLaunchKernelGGL((k_gemm_split_p<false, 2, 4, 3>), dim3(std::min(g.n_tiles, cap)), dim3(1024), 0, 0, g);
        else if (NP == 2) hipLaunchKernelGGL((k_gemm_split_p<false, 1, 2, 2>), dim3(std::min(g.n_tiles, cap)), dim3(1024), 0, 0, g);
        else hipLaunchKernelGGL((k_gemm_split_p<false, 1, 2, 3>), dim3(std::min(g.n_tiles, cap)), dim3(1024), 0, 0, g);
    };
    launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    float ms;
    const int sleep_us = getenv("GS_SLEEP_US") ? atoi(getenv("GS_SLEEP_US")) : 0;
    if (sleep_us > 0) {
        std::vector<float> t;
        for (int i = 0; i < reps; ++i) {
            usleep(sleep_us);
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float m1; CK(hipEventElapsedTime(&m1, e0, e1)); t.push_back(m1);
        }
        std::sort(t.begin(), t.end()); ms = t[t.size() / 2];
        printf("cool-chip mode (pause %d us): min %.3f median %.3f max %.3f ms\n", sleep_us, t.front(), ms, t.back());
    } else {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    }
    const double fl = 2.0 * M * K * N;
    printf("NP=%d GEMM: %.3f ms  %.1f TFLOP/s (fp32-equivalent)  %.2f TB/s of A+C traffic\n", NP, ms, fl / ms / 1e9, ((double)M * K * 4 + (double)M * N * 4) / ms / 1e9);
    std::vector<float> C((size_t)M * N);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    unsigned hmax[3]; CK(hipMemcpy(hmax, dMax, 12, hipMemcpyDeviceToHost));
    float fm[3]; for (int i = 0; i < 3; ++i) { union { unsigned u; float f; } c; c.u = hmax[i]; fm[i] = c.f; }
    // accuracy on a sample of rows, relative to each row's own output scale: vs fp64, next to the fp32 fmaf chain's own error
    double worst_split = 0, worst_f32 = 0, rms_split = 0, rms_f32 = 0, cmax = 0; size_t cnt = 0;
    for (int s = 0; s < 512; ++s) {
        const int row = (int)(((int64_t)s * 7919 * 131) % M);
        double rs = 0; std::vector<double> ref(N); std::vector<float> f32(N);
        for (int n = 0; n < N; ++n) {
            double r = 0; float f = 0.f;
            for (int k = 0; k < K; ++k) { r += (double)A[(size_t)row * K + k] * (double)W[(size_t)k * N + n]; f = fmaf(A[(size_t)row * K + k], W[(size_t)k * N + n], f); }
            ref[n] = r + bias[n]; f32[n] = f + bias[n]; rs += ref[n] * ref[n];
        }
        rs = sqrt(rs / N);
        for (int n = 0; n < N; ++n) {
            const double es = fabs((double)C[(size_t)row * N + n] - ref[n]) / rs, ef = fabs((double)f32[n] - ref[n]) / rs;
            worst_split = fmax(worst_split, es); worst_f32 = fmax(worst_f32, ef); rms_split += es * es; rms_f32 += ef * ef; ++cnt;
            cmax = fmax(cmax, fabs((double)C[(size_t)row * N + n]));
        }
    }
    printf("error / row rms of C over 512 rows:  this kernel max %.3e rms %.3e   fp32 fmaf chain max %.3e rms %.3e\n", worst_split, sqrt(rms_split / cnt), worst_f32, sqrt(rms_f32 / cnt));
    printf("recorded bounds: A %.4g  W %.4g  C %.4g (sampled max |C| %.4g)\n", fm[0], fm[1], fm[2], cmax);
    return worst_split < 4 * worst_f32 + 1e-7 ? 0 : 2;
}
