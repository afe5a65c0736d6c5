// Streaming-copy ceiling of one MI355X for the access idioms the aggregate kernel uses (1-KiB row per wave, float4 per lane).
//   hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/_build/membw && tools/_build/membw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));

template <int NT_LD, int NT_ST, int UNR>
__global__ __launch_bounds__(256) void k_copy(const f4v* __restrict__ in, f4v* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x;
    f4v v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (i + u * 256 < n4) v[u] = NT_LD ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (i + u * 256 < n4) { if (NT_ST) __builtin_nontemporal_store(v[u], out + i + u * 256); else out[i + u * 256] = v[u]; }
}
// rows of 1 KiB permuted within windows of W rows (gather like the aggregate: a wave reads one whole row per access)
template <int NT_ST, int UNR>
__global__ __launch_bounds__(256) void k_rowcopy(const f4v* __restrict__ in, f4v* __restrict__ out, const int* __restrict__ perm, size_t rows) {
    const int lane = threadIdx.x & 63;
    size_t r0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * UNR;
    f4v v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (r0 + u < rows) v[u] = in[(size_t)perm[r0 + u] * 64 + lane];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (r0 + u < rows) { if (NT_ST) __builtin_nontemporal_store(v[u], out + (r0 + u) * 64 + lane); else out[(r0 + u) * 64 + lane] = v[u]; }
}
// column-chunk copy: the matrix [rows, 256 floats] is copied one CW-float column chunk at a time (blockIdx.y = chunk):
// each access of a wave covers 64*16/(CW*4) rows x CW*4 contiguous bytes at a 1-KiB stride (what an LDS-staged
// per-subgraph aggregate would issue)
template <int CW, int UNR, int NT_ST>
__global__ __launch_bounds__(256) void k_chunkcopy(const f4v* __restrict__ in, f4v* __restrict__ out, size_t rows) {
    constexpr int LPR = CW / 4, RPW = 64 / LPR;                 // lanes per row segment, rows per wave access
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * RPW * UNR + lane / LPR;
    const size_t col = (size_t)blockIdx.y * LPR + lane % LPR;
    f4v v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (r0 + u * RPW < rows) v[u] = in[(r0 + u * RPW) * 64 + col];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (r0 + u * RPW < rows) { if (NT_ST) __builtin_nontemporal_store(v[u], out + (r0 + u * RPW) * 64 + col); else out[(r0 + u * RPW) * 64 + col] = v[u]; }
}
// LDS-staged per-subgraph copy: one 1024-thread workgroup per block of SUB rows walks the row's 256 floats in CW-float
// column chunks: chunk c+1 is loaded into registers while chunk c is served from LDS (rows permuted = the gather) and
// written out.  The shape an LDS-staged per-subgraph aggregate would have.
template <int CW, int SUB, int NT_ST>
__global__ __launch_bounds__(1024) void k_subcopy(const f4v* __restrict__ in, f4v* __restrict__ out, const int* __restrict__ perm, size_t rows) {
    constexpr int LPR = CW / 4, PER = SUB * LPR, PF = (PER + 1023) / 1024, NCH = 256 / CW;
    extern __shared__ f4v lds[];                       // 2 buffers of SUB * LPR float4
    const size_t r0 = (size_t)blockIdx.x * SUB;
    const int tid = threadIdx.x;
    f4v pf[PF];
    auto load = [&](int c) {
#pragma unroll
        for (int p = 0; p < PF; ++p) { const int id = tid + p * 1024; const int r = id / LPR, l = id % LPR; if (id < PER && r0 + r < rows) pf[p] = in[(r0 + r) * 64 + c * LPR + l]; }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int p = 0; p < PF; ++p) { const int id = tid + p * 1024; if (id < PER) lds[b * PER + id] = pf[p]; }
    };
    load(0); stash(0); __syncthreads();
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) load(c + 1);
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int id = tid + p * 1024; const int r = id / LPR, l = id % LPR;
            if (id < PER && r0 + r < rows) {
                const int src = perm[r0 + r] - (int)r0;                 // permuted inside the block
                const f4v v = lds[(c & 1) * PER + src * LPR + l];
                if (NT_ST) __builtin_nontemporal_store(v, out + (r0 + r) * 64 + c * LPR + l); else out[(r0 + r) * 64 + c * LPR + l] = v;
            }
        }
        if (c + 1 < NCH) stash((c + 1) & 1);
        __syncthreads();
    }
}
template <int UNR>
__global__ __launch_bounds__(256) void k_read(const f4v* __restrict__ in, float* sink, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x;
    f4v a = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (i + u * 256 < n4) a += in[i + u * 256];
    if (a.x + a.y + a.z + a.w == 123.456f) *sink = 1.f;
}
template <int NT_ST, int UNR>
__global__ __launch_bounds__(256) void k_write(f4v* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x;
    f4v v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (i + u * 256 < n4) { if (NT_ST) __builtin_nontemporal_store(v, out + i + u * 256); else out[i + u * 256] = v; }
}

static hipEvent_t e0, e1;
template <class F>
static void timeit(const char* name, double bytes, F launch) {
    for (int w = 0; w < 3; ++w) launch();
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us  %7.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
}
#define TIME(name, bytes, launch) timeit(name, (double)(bytes), [&]() { launch; })

int main(int argc, char** argv) {
    const size_t MB = argc > 1 ? atol(argv[1]) : 1172;       // one side; default = the arxiv query batch at F=256
    const size_t n4 = MB * 1024 * 1024 / 16, rows = n4 / 64;
    f4v *a, *b; float* sink; int* perm;
    hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&sink, 4); hipMalloc(&perm, rows * 4);
    hipMemset(a, 0, n4 * 16); hipMemset(b, 0, n4 * 16);
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("buffer %zu MB per side, %zu rows of 1 KiB\n", MB, rows);
#define G(U) dim3((unsigned)((n4 + 256 * U - 1) / (256 * U)))
    TIME("copy ld/st                 unr1", 2 * n4 * 16, hipLaunchKernelGGL((k_copy<0, 0, 1>), G(1), dim3(256), 0, 0, a, b, n4));
    TIME("copy ld/st                 unr4", 2 * n4 * 16, hipLaunchKernelGGL((k_copy<0, 0, 4>), G(4), dim3(256), 0, 0, a, b, n4));
    TIME("copy ld/st-nt              unr1", 2 * n4 * 16, hipLaunchKernelGGL((k_copy<0, 1, 1>), G(1), dim3(256), 0, 0, a, b, n4));
    TIME("copy ld/st-nt              unr4", 2 * n4 * 16, hipLaunchKernelGGL((k_copy<0, 1, 4>), G(4), dim3(256), 0, 0, a, b, n4));
    TIME("copy ld-nt/st-nt           unr4", 2 * n4 * 16, hipLaunchKernelGGL((k_copy<1, 1, 4>), G(4), dim3(256), 0, 0, a, b, n4));
    TIME("copy ld-nt/st-nt           unr8", 2 * n4 * 16, hipLaunchKernelGGL((k_copy<1, 1, 8>), G(8), dim3(256), 0, 0, a, b, n4));
    TIME("read only                  unr4", n4 * 16, hipLaunchKernelGGL((k_read<4>), G(4), dim3(256), 0, 0, a, sink, n4));
    TIME("write only                 unr4", n4 * 16, hipLaunchKernelGGL((k_write<0, 4>), G(4), dim3(256), 0, 0, b, n4));
    TIME("write only nt              unr4", n4 * 16, hipLaunchKernelGGL((k_write<1, 4>), G(4), dim3(256), 0, 0, b, n4));
#define GC(CW, U) dim3((unsigned)((rows + 4 * (64 / (CW / 4)) * U - 1) / (4 * (64 / (CW / 4)) * U)), 256 / CW)
    TIME("chunk copy 256-B segments  unr2 st-nt", 2 * n4 * 16, hipLaunchKernelGGL((k_chunkcopy<64, 2, 1>), GC(64, 2), dim3(256), 0, 0, a, b, rows));
    TIME("chunk copy 256-B segments  unr4 st-nt", 2 * n4 * 16, hipLaunchKernelGGL((k_chunkcopy<64, 4, 1>), GC(64, 4), dim3(256), 0, 0, a, b, rows));
    TIME("chunk copy 256-B segments  unr4 st", 2 * n4 * 16, hipLaunchKernelGGL((k_chunkcopy<64, 4, 0>), GC(64, 4), dim3(256), 0, 0, a, b, rows));
    TIME("chunk copy 128-B segments  unr4 st-nt", 2 * n4 * 16, hipLaunchKernelGGL((k_chunkcopy<32, 4, 1>), GC(32, 4), dim3(256), 0, 0, a, b, rows));
    TIME("chunk copy 512-B segments  unr2 st-nt", 2 * n4 * 16, hipLaunchKernelGGL((k_chunkcopy<128, 2, 1>), GC(128, 2), dim3(256), 0, 0, a, b, rows));
    {   // rows permuted inside blocks of 512 rows for the LDS-staged variants
        std::vector<int> h(rows);
        for (size_t r = 0; r < rows; ++r) h[r] = (int)r;
        srand(1); for (size_t w0 = 0; w0 + 512 <= rows; w0 += 512) for (int i = 511; i > 0; --i) { int j = rand() % (i + 1); std::swap(h[w0 + i], h[w0 + j]); }
        hipMemcpy(perm, h.data(), rows * 4, hipMemcpyHostToDevice);
        const unsigned nb = (unsigned)((rows + 511) / 512);
        hipFuncSetAttribute((const void*)k_subcopy<64, 512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_subcopy<32, 512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_subcopy<32, 512, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_subcopy<16, 512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        TIME("LDS-staged sub copy CW=64 (256-B) 512 rows", 2 * n4 * 16, hipLaunchKernelGGL((k_subcopy<64, 512, 1>), dim3(nb), dim3(1024), 2 * 512 * 16 * 16, 0, a, b, perm, rows));
        TIME("LDS-staged sub copy CW=32 (128-B) 512 rows", 2 * n4 * 16, hipLaunchKernelGGL((k_subcopy<32, 512, 1>), dim3(nb), dim3(1024), 2 * 512 * 8 * 16, 0, a, b, perm, rows));
        TIME("LDS-staged sub copy CW=32 st (no nt)      ", 2 * n4 * 16, hipLaunchKernelGGL((k_subcopy<32, 512, 0>), dim3(nb), dim3(1024), 2 * 512 * 8 * 16, 0, a, b, perm, rows));
        TIME("LDS-staged sub copy CW=16 (64-B)  512 rows", 2 * n4 * 16, hipLaunchKernelGGL((k_subcopy<16, 512, 1>), dim3(nb), dim3(1024), 2 * 512 * 4 * 16, 0, a, b, perm, rows));
    }
    for (int W : {1, 512, 4096}) {          // W = 1: identity; else rows shuffled inside windows of W rows (subgraph-local gather)
        std::vector<int> h(rows);
        for (size_t r = 0; r < rows; ++r) h[r] = (int)r;
        if (W > 1) { srand(1); for (size_t w0 = 0; w0 + W <= rows; w0 += W) for (int i = W - 1; i > 0; --i) { int j = rand() % (i + 1); std::swap(h[w0 + i], h[w0 + j]); } }
        hipMemcpy(perm, h.data(), rows * 4, hipMemcpyHostToDevice);
        char nm[64];
#define GR(U) dim3((unsigned)((rows + 4 * U - 1) / (4 * U)))
        snprintf(nm, 64, "row gather W=%-5d st       unr2", W); TIME(nm, 2 * n4 * 16, hipLaunchKernelGGL((k_rowcopy<0, 2>), GR(2), dim3(256), 0, 0, a, b, perm, rows));
        snprintf(nm, 64, "row gather W=%-5d st-nt    unr2", W); TIME(nm, 2 * n4 * 16, hipLaunchKernelGGL((k_rowcopy<1, 2>), GR(2), dim3(256), 0, 0, a, b, perm, rows));
        snprintf(nm, 64, "row gather W=%-5d st-nt    unr8", W); TIME(nm, 2 * n4 * 16, hipLaunchKernelGGL((k_rowcopy<1, 8>), GR(8), dim3(256), 0, 0, a, b, perm, rows));
    }
    return 0;
}
