// Write-throughput probe for the NT GEMM epilogue's store pattern (tools/probes: measurement only, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_patterns tools/probes/store_patterns.hip && /tmp/store_patterns
// C [M, 750] fp32 at pitch ld.  Patterns, one workgroup of 512 threads per 256 x 256 tile unless noted:
//   0  the nt64 epilogue: wave w owns 32 columns, an instruction = 8 rows x 128 bytes
//   1  workgroup-cooperative: an instruction = one row x 1 KB (wave w: rows w, w + 8, ...)
//   2  whole rows: a workgroup of 256 threads per 8 consecutive rows x all columns (bn_act_fwd_rowseg's shape)
//   3  pattern 0 with tiles 256 rows x 768 columns walked column tile by column tile by ONE workgroup
// `lds` bytes of dynamic LDS limit the workgroups per CU (the GEMM holds 128 KB: one per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(512) void pat(float* C, int M, int N, long ld, int tiles_n, int pattern, int vec4) {
    extern __shared__ float dummy[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    int tm, tn0, tn1;
    if (pattern == 3) tm = j * 8 + xcd, tn0 = 0, tn1 = tiles_n;
    else tm = (j / tiles_n) * 8 + xcd, tn0 = j % tiles_n, tn1 = tn0 + 1;
    const int m0 = tm * 256;
    if (m0 >= M) return;
    for (int tn = tn0; tn < tn1; ++tn) {
        const int c0 = tn * 256;
        if (pattern == 0 || pattern == 3) {
            const int col = c0 + w * 32 + (lane & 7) * 4;
            for (int pass = 0; pass < 8; ++pass)
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + pass * 32 + i * 8 + (lane >> 3);
                    if (row < M && col + 3 < N) {
                        float* c = C + (long)row * ld + col;
                        if (vec4) *reinterpret_cast<float4*>(c) = make_float4(1.f, 2.f, 3.f, 4.f);
                        else *reinterpret_cast<float2*>(c) = make_float2(1.f, 2.f), *reinterpret_cast<float2*>(c + 2) = make_float2(3.f, 4.f);
                    }
                }
        } else {
            const int col = c0 + lane * 4;
            for (int pass = 0; pass < 8; ++pass)
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + pass * 32 + i * 8 + w;
                    if (row < M && col + 3 < N) {
                        float* c = C + (long)row * ld + col;
                        if (vec4) *reinterpret_cast<float4*>(c) = make_float4(1.f, 2.f, 3.f, 4.f);
                        else *reinterpret_cast<float2*>(c) = make_float2(1.f, 2.f), *reinterpret_cast<float2*>(c + 2) = make_float2(3.f, 4.f);
                    }
                }
        }
    }
}

__global__ __launch_bounds__(256) void rows(float* C, int M, int N, long ld, int vec4) {
    const int col = threadIdx.x * 4;
    if (col + 3 >= N) return;
    for (int u = 0; u < 8; ++u) {
        const long row = (long)blockIdx.x * 8 + u;
        if (row >= M) break;
        float* c = C + row * ld + col;
        if (vec4) *reinterpret_cast<float4*>(c) = make_float4(1.f, 2.f, 3.f, 4.f);
        else *reinterpret_cast<float2*>(c) = make_float2(1.f, 2.f), *reinterpret_cast<float2*>(c + 2) = make_float2(3.f, 4.f);
    }
}

int main() {
    const int M = 169343, N = 748;      // (748: whole quads)
    float* C;
    hipMalloc(&C, (size_t)M * 752 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    for (long ld : {750L, 752L})
        for (int lds : {0, 131072})
            for (int pattern = 0; pattern < 4; ++pattern) {
                if (pattern == 2 && lds) continue;
                const int vec4 = ld % 4 == 0;
                float best = 1e9;
                for (int rep = 0; rep < 6; ++rep) {
                    hipEventRecord(a);
                    if (pattern == 2) hipLaunchKernelGGL(rows, dim3((M + 7) / 8), dim3(256), 0, 0, C, M, N, ld, vec4);
                    else if (pattern == 3) hipLaunchKernelGGL(pat, dim3((tiles_m + 7) / 8 * 8), dim3(512), lds, 0, C, M, N, ld, tiles_n, pattern, vec4);
                    else hipLaunchKernelGGL(pat, dim3((tiles_m + 7) / 8 * 8 * tiles_n), dim3(512), lds, 0, C, M, N, ld, tiles_n, pattern, vec4);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms;
                    hipEventElapsedTime(&ms, a, b);
                    if (rep && ms < best) best = ms;
                }
                printf("ld=%ld lds=%6d pattern %d: %.1f us  %.2f TB/s\n", ld, lds, pattern, best * 1e3, (double)M * N * 4 / best / 1e9);
            }
    return 0;
}
