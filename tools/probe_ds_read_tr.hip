#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];   // [row][col], 64 cols
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)((i / 64) * 256 + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x, t = lane & 15, G = lane >> 4;
    // candidate addressing: lane t of a group reads row (t >> 2) ... 4 halves at col 4 * (t & 3); group G offsets rows by 4 * G
    const int row = 4 * G + (t >> 2), col = 4 * (t & 3);
#if defined(__HIP_DEVICE_COMPILE__)
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + row * 64 + col));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
#endif
}
int main() {
    int* d; hipMalloc(&d, 256 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l*4+j] >> 8, h[l*4+j] & 255); printf("\n"); }
    return 0;
}
