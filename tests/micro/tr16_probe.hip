// tr16_probe.hip -- developer probe (not product code): lane mapping and bank behaviour of ds_read_b64_tr_b16 on gfx950.
//   part 1: LDS holds lds[i] = i (16-bit); lane l reads the 8 bytes at element 4*l.  Prints result[lane][0..3] and checks
//           the mapping gemm_bf16_tn.hpp assumes: within a 16-lane group the 16 x 4 elements are a row-major [4][16] block
//           (source lane s holds row s/4, columns 4*(s%4)..+3) and result lane c receives column c: rows 0..3.
//   part 2: cycles per fragment read of a [64 k][128 n] bf16 tile (256-byte rows) for the MFMA 32x32x16 operand pattern,
//           with and without the row swizzle (16-byte slot ^= 4 * (k & 3)), against plain ds_read_b128 of a K-contiguous tile.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
#define LDS3(p) ((__attribute__((address_space(3))) v4s*)(p))

__global__ void probe_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}

// MODE 0: tr reads, no swizzle; 1: tr reads, slot ^= 4*(k&3); 2: plain b128 from a [128 n][64 k] image with the GEMM's swizzle
template <int MODE>
__global__ __launch_bounds__(512) void bank_kernel(long long* cycles, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384 / 4; i += 512) reinterpret_cast<int*>(sm)[i] = i;
    __syncthreads();
    const int g = lane >> 4, s = lane & 15;
    const int nblk = (wave & 3) * 32;     // this wave's 32 columns
    int acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {      // four k-steps of 16
            if (MODE == 2) {
                const int row = nblk + (lane & 31), h = lane >> 5;
                const int swz = ((row >> 1) ^ (row >> 4)) & 7;
                const v8s v = *reinterpret_cast<const v8s*>(sm + row * 128 + 16 * ((2 * t + h) ^ swz));
                acc ^= v[0] ^ v[7];
            } else {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int k = 16 * t + 8 * (g >> 1) + 4 * r + (s >> 2);
                    int slot = (nblk + 16 * (g & 1) + 4 * (s & 3)) >> 3;          // 16-byte slot of the row (8 elements)
                    const int half = ((nblk + 16 * (g & 1) + 4 * (s & 3)) >> 2) & 1; // 8-byte half of the slot
                    if (MODE == 1) slot ^= 4 * (k & 3);
                    const v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(sm + k * 256 + slot * 16 + half * 8));
                    acc ^= v[0] ^ v[3];
                }
            }
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc == 0x7fffffff) *sink = acc;
}

int main() {
    short* out;
    CK(hipMalloc(&out, 256 * 2));
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, out);
    short h[256];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    int ok = 1;
    printf("lane: result elements (as source element index = 4 * source lane + e)\n");
    for (int l = 0; l < 64; ++l) {
        printf("%2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
        const int g = l >> 4, c = l & 15;
        for (int j = 0; j < 4; ++j) {
            const int want = 4 * (16 * g + 4 * j + (c >> 2)) + (c & 3);
            if (h[4 * l + j] != want) ok = 0;
        }
    }
    printf("TR16_MAPPING_%s  (assumed: [4 rows][16 cols] block per 16 lanes -> lane c gets column c)\n", ok ? "AS_ASSUMED" : "DIFFERENT");
    long long* cyc;
    int* sink;
    CK(hipMalloc(&cyc, 8 * 8 * 8));
    CK(hipMalloc(&sink, 4));
    const int iters = 2000;
    const char* names[] = {"tr16 b64, linear rows", "tr16 b64, slot ^= 4*(k&3)", "b128 K-contiguous (GEMM today)"};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(bank_kernel<0>, dim3(1), dim3(512), 16384, 0, cyc, iters, sink);
            if (m == 1) hipLaunchKernelGGL(bank_kernel<1>, dim3(1), dim3(512), 16384, 0, cyc, iters, sink);
            if (m == 2) hipLaunchKernelGGL(bank_kernel<2>, dim3(1), dim3(512), 16384, 0, cyc, iters, sink);
        }
        long long hc[8];
        CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
        double mx = 0;
        for (int w = 0; w < 8; ++w) mx = hc[w] > mx ? hc[w] : mx;
        printf("bank %d (%s): %.1f cycles per K-tile fragment set (8 waves, 1 KiB per wave: 8 tr reads or 4 b128 reads)\n", m, names[m], mx / iters);
    }
    return 0;
}
