// Micro-test of the sweep kernel's prefetch ring protocol on gfx950: per round 2 x 16-byte stores then
// 2 x 16-byte asm loads, ring depth 8, steady-state wait vmcnt(28), first lap vmcnt(14 + 2 i).
// Every consumed value is verified.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned expect(size_t i) { return (unsigned)(i * 2654435761u) ^ 0x5bd1e995u; }
__global__ void __launch_bounds__(256) ring(const unsigned* a, const unsigned* b, double* out, unsigned long long* bad, int Q, size_t per_block) {
    const int tid = threadIdx.x, T = blockDim.x;
    const unsigned* A = a + blockIdx.x * per_block;
    const unsigned* B = b + blockIdx.x * per_block;
    double* O = out + blockIdx.x * per_block;
    u32x4 rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7, rn0, rn1, rn2, rn3, rn4, rn5, rn6, rn7;
    unsigned long long nbad = 0;
#define RING_LOAD(i, q) { int g_ = (q) * T + tid; g_ = g_ > Q * T - 1 ? Q * T - 1 : g_; \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off" : "=&v"(rc##i), "=&v"(rn##i) : "v"(A + 4 * g_), "v"(B + 4 * g_) : "memory"); }
#define RING_WAIT(i, n) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(rc##i), "+v"(rn##i)::"memory");
#define ROUND(i, qq, n) RING_WAIT(i, n) { \
    const size_t p0 = 4 * ((size_t)(qq) * T + tid); \
    if ((qq) < Q) { _Pragma("unroll") for (int j = 0; j < 4; j++) { \
        if (rc##i[j] != expect(blockIdx.x * per_block + p0 + j)) nbad++; \
        if (rn##i[j] != (expect(blockIdx.x * per_block + p0 + j) ^ 0xFFFFu)) nbad++; } } \
    double* dst = (qq) < Q ? O + p0 : O; \
    f64x2 lo = {(double) rc##i[0], (double) rc##i[1]}, hi = {(double) rn##i[2], (double) rn##i[3]}; \
    *reinterpret_cast<f64x2*>(dst) = lo; *reinterpret_cast<f64x2*>(dst + 2) = hi; } \
    RING_LOAD(i, (qq) + 8)
    RING_LOAD(0,0) RING_LOAD(1,1) RING_LOAD(2,2) RING_LOAD(3,3) RING_LOAD(4,4) RING_LOAD(5,5) RING_LOAD(6,6) RING_LOAD(7,7)
    ROUND(0,0,14) ROUND(1,1,16) ROUND(2,2,18) ROUND(3,3,20) ROUND(4,4,22) ROUND(5,5,24) ROUND(6,6,26) ROUND(7,7,28)
    for (int q0 = 8; q0 < Q; q0 += 8) {
        ROUND(0,q0,28) ROUND(1,q0+1,28) ROUND(2,q0+2,28) ROUND(3,q0+3,28) ROUND(4,q0+4,28) ROUND(5,q0+5,28) ROUND(6,q0+6,28) ROUND(7,q0+7,28)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nbad) atomicAdd(bad, nbad);
}
__global__ void fill(unsigned* a, unsigned* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { a[i] = expect(i); b[i] = expect(i) ^ 0xFFFFu; }
}
int main() {
    const int blocks = 2048, T = 256, Q = 96;
    const size_t per_block = (size_t) Q * T * 4, n = per_block * blocks;
    unsigned *a, *b; double* out; unsigned long long* bad;
    hipMalloc(&a, n * 4 + 64); hipMalloc(&b, n * 4 + 64); hipMalloc(&out, n * 8 + 64); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    fill<<<4096, 256>>>(a, b, n);
    for (int rep = 0; rep < 20; rep++) ring<<<blocks, T>>>(a, b, out, bad, Q, per_block);
    unsigned long long h = 0; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("elements checked %.3e x20, bad %llu\n", (double) n * 2, h);
    return 0;
}
