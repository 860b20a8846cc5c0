// Micro-test of the cross-lane primitives the prune kernel's single-wave section uses on gfx950:
// DPP inclusive scan, DPP-based xor exchanges for strides 1..8, and a 128-key bitonic sort built on them,
// each against the portable __shfl version.  Prints the number of mismatches (expected 0) and cycle counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define WAVE 64
template <int CTRL, int ROWMASK = 0xf>
static __device__ __forceinline__ uint32_t dpp_mov(uint32_t x) {
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, CTRL, ROWMASK, 0xf, false);
}
static __device__ __forceinline__ int scan_dpp(int v, int lane) {
    int t;
    t = (int) dpp_mov<0x111>((uint32_t) v) + v; if ((lane & 15) >= 1) v = t;
    t = (int) dpp_mov<0x112>((uint32_t) v) + v; if ((lane & 15) >= 2) v = t;
    t = (int) dpp_mov<0x114>((uint32_t) v) + v; if ((lane & 15) >= 4) v = t;
    t = (int) dpp_mov<0x118>((uint32_t) v) + v; if ((lane & 15) >= 8) v = t;
    t = (int) dpp_mov<0x142, 0xa>((uint32_t) v) + v; if ((lane & 31) >= 16) v = t;
    t = (int) dpp_mov<0x143, 0xc>((uint32_t) v) + v; if (lane >= 32) v = t;
    return v;
}
static __device__ __forceinline__ int scan_shfl(int v, int lane) {
    for (int o = 1; o < WAVE; o <<= 1) { const int t = __shfl_up(v, o, WAVE); if (lane >= o) v += t; }
    return v;
}
template <int J>
static __device__ __forceinline__ uint32_t xor_dpp(uint32_t x, int lane) {
    if (J == 1) return dpp_mov<0xB1>(x);            // quad_perm [1,0,3,2]
    if (J == 2) return dpp_mov<0x4E>(x);            // quad_perm [2,3,0,1]
    if (J == 4) { const uint32_t a = dpp_mov<0x104>(x), b = dpp_mov<0x114>(x); return (lane & 4) ? b : a; }  // row_shl:4 / row_shr:4
    if (J == 8) { const uint32_t a = dpp_mov<0x108>(x), b = dpp_mov<0x118>(x); return (lane & 8) ? b : a; }
    return (uint32_t) __shfl_xor((int) x, J, WAVE);
}
template <bool DPP>
static __device__ __forceinline__ void bitonic128(uint32_t &k0, uint32_t &k1, int lane) {
#pragma unroll
    for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == 64) { const uint32_t lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0; k0 = lo; k1 = hi; }
            else {
                uint32_t p0, p1;
                if (DPP && j == 1) { p0 = xor_dpp<1>(k0, lane); p1 = xor_dpp<1>(k1, lane); }
                else if (DPP && j == 2) { p0 = xor_dpp<2>(k0, lane); p1 = xor_dpp<2>(k1, lane); }
                else if (DPP && j == 4) { p0 = xor_dpp<4>(k0, lane); p1 = xor_dpp<4>(k1, lane); }
                else if (DPP && j == 8) { p0 = xor_dpp<8>(k0, lane); p1 = xor_dpp<8>(k1, lane); }
                else { p0 = (uint32_t) __shfl_xor((int) k0, j, WAVE); p1 = (uint32_t) __shfl_xor((int) k1, j, WAVE); }
                const bool lower = (lane & j) == 0;
                const bool asc0 = (lane & k) == 0, asc1 = ((lane + 64) & k) == 0;
                const uint32_t mn0 = k0 < p0 ? k0 : p0, mx0 = k0 < p0 ? p0 : k0, mn1 = k1 < p1 ? k1 : p1, mx1 = k1 < p1 ? p1 : k1;
                k0 = (lower == asc0) ? mn0 : mx0;
                k1 = (lower == asc1) ? mn1 : mx1;
            }
        }
    }
}
__global__ void probe(const uint32_t *in, uint32_t *out, unsigned long long *cyc) {
    const int lane = threadIdx.x;
    const uint32_t *src = in + blockIdx.x * 128;
    uint32_t *dst = out + blockIdx.x * 512;
    const int v = (int) (src[lane] & 0xFFFF);
    dst[lane] = (uint32_t) scan_dpp(v, lane);
    dst[64 + lane] = (uint32_t) scan_shfl(v, lane);
    uint32_t a0 = src[lane], a1 = src[64 + lane], b0 = a0, b1 = a1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bitonic128<true>(a0, a1, lane);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    bitonic128<false>(b0, b1, lane);
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    dst[128 + lane] = a0; dst[192 + lane] = a1; dst[256 + lane] = b0; dst[320 + lane] = b1;
    dst[384 + lane] = xor_dpp<4>(src[lane], lane) ^ (uint32_t) __shfl_xor((int) src[lane], 4, WAVE);
    dst[448 + lane] = xor_dpp<8>(src[lane], lane) ^ (uint32_t) __shfl_xor((int) src[lane], 8, WAVE);
    if (lane == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
template <bool DPP>
__global__ void sort_loop(const uint32_t *in, uint32_t *out, int iters) {
    const int lane = threadIdx.x;
    uint32_t a0 = in[lane], a1 = in[64 + lane];
    for (int i = 0; i < iters; i++) {
        bitonic128<DPP>(a0, a1, lane);
        a0 = a0 * 2654435761u + (uint32_t) i; a1 = a1 * 2246822519u + (uint32_t) lane; /* unsorted again, serial dependency */
    }
    out[lane] = a0; out[64 + lane] = a1;
}
template <bool DPP>
__global__ void scan_loop(const uint32_t *in, uint32_t *out, int iters) {
    const int lane = threadIdx.x;
    int v = (int) (in[lane] & 0xFF);
    for (int i = 0; i < iters; i++) v = (DPP ? scan_dpp(v, lane) : scan_shfl(v, lane)) & 0xFF;
    out[lane] = (uint32_t) v;
}
template <class K>
static float time_kernel(K launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const int B = 256;
    std::vector<uint32_t> h(B * 128);
    srand(1);
    for (auto &x : h) x = (uint32_t) rand() * 2654435761u;
    uint32_t *din, *dout; unsigned long long *dc;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, B * 512 * 4); hipMalloc(&dc, 16);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(B), dim3(64), 0, 0, din, dout, dc);
    std::vector<uint32_t> o(B * 512); unsigned long long c[2];
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    long bad_scan = 0, bad_sort_dpp = 0, bad_sort_shfl = 0, bad_x = 0;
    for (int b = 0; b < B; b++) {
        int acc = 0;
        for (int l = 0; l < 64; l++) { acc += (int) (h[b * 128 + l] & 0xFFFF); bad_scan += o[b * 512 + l] != (uint32_t) acc; bad_scan += o[b * 512 + 64 + l] != (uint32_t) acc; }
        std::vector<uint32_t> s(h.begin() + b * 128, h.begin() + b * 128 + 128);
        std::sort(s.begin(), s.end());
        for (int i = 0; i < 128; i++) { bad_sort_dpp += o[b * 512 + 128 + i] != s[i]; bad_sort_shfl += o[b * 512 + 256 + i] != s[i]; }
        for (int l = 0; l < 128; l++) bad_x += o[b * 512 + 384 + l] != 0;
    }
    printf("mismatches: scan %ld, bitonic(dpp) %ld, bitonic(shfl) %ld, xor4/8 %ld; cycles bitonic dpp %llu shfl %llu\n", bad_scan, bad_sort_dpp,
           bad_sort_shfl, bad_x, c[0], c[1]);
    const int iters = 2000;
    const float s_dpp = time_kernel([&] { hipLaunchKernelGGL(sort_loop<true>, dim3(1), dim3(64), 0, 0, din, dout, iters); });
    const float s_shf = time_kernel([&] { hipLaunchKernelGGL(sort_loop<false>, dim3(1), dim3(64), 0, 0, din, dout, iters); });
    const float c_dpp = time_kernel([&] { hipLaunchKernelGGL(scan_loop<true>, dim3(1), dim3(64), 0, 0, din, dout, iters); });
    const float c_shf = time_kernel([&] { hipLaunchKernelGGL(scan_loop<false>, dim3(1), dim3(64), 0, 0, din, dout, iters); });
    printf("one wave, dependent chain: bitonic128 dpp %.3f us, shfl %.3f us; scan64 dpp %.3f us, shfl %.3f us\n", 1e3 * s_dpp / iters,
           1e3 * s_shf / iters, 1e3 * c_dpp / iters, 1e3 * c_shf / iters);
    return (bad_scan || bad_sort_dpp || bad_sort_shfl || bad_x) ? 1 : 0;
}
