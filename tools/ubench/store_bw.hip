// Store-bandwidth microbenchmark (MI355X): how fast can fresh data be written, by access shape and alignment?
// build: hipcc -O3 --offload-arch=gfx950 -o store_bw store_bw.hip ; run: ./store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed, aligned(4))) u2 { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) u4 { uint32_t x, y, z, w; };
template <int MODE>
__global__ void k(uint32_t *p, size_t n_dw, int off) {
    // each wave writes contiguous runs; grid-stride
    size_t tid = (size_t) blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t) gridDim.x * blockDim.x;
    if (MODE == 0) for (size_t i = tid; i < n_dw; i += nt) p[i + off] = (uint32_t) i;                                    // dword
    if (MODE == 1) for (size_t i = tid; i < n_dw / 2; i += nt) { u2 v = {(uint32_t) i, 1u}; *(u2 *) (p + 2 * i + off) = v; } // 8 B, offset off dwords
    if (MODE == 2) for (size_t i = tid; i < n_dw / 4; i += nt) { u4 v = {(uint32_t) i, 1u, 2u, 3u}; *(u4 *) (p + 4 * i + off) = v; } // 16 B
    if (MODE == 3) { // two arrays, 8 B each per lane, rows of 408 B starting anywhere (the cross product's shape)
        uint32_t *q = p + n_dw / 2;
        for (size_t i = tid; i < n_dw / 4; i += nt) { u2 v = {(uint32_t) i, 1u}; *(u2 *) (p + 2 * i + off) = v; *(u2 *) (q + 2 * i + off) = v; }
    }
}
int main() {
    const size_t n_dw = (size_t) 1 << 30; // 4 GiB
    uint32_t *p; hipMalloc(&p, n_dw * 4 + 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto kern, int off, int grid) {
        kern<<<grid, 256>>>(p, n_dw, off); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 3; r++) kern<<<grid, 256>>>(p, n_dw, off); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
        printf("%-28s off %d grid %6d: %7.3f ms  %6.2f TB/s\n", name, off, grid, ms, n_dw * 4 / ms / 1e9);
    };
    for (int grid : {2048, 16384, 131072}) {
        run("dword", k<0>, 0, grid); run("dword", k<0>, 1, grid);
        run("8B", k<1>, 0, grid); run("8B", k<1>, 1, grid);
        run("16B", k<2>, 0, grid); run("16B", k<2>, 1, grid); run("16B", k<2>, 2, grid);
        run("2 arrays x 8B", k<3>, 0, grid); run("2 arrays x 8B", k<3>, 1, grid);
    }
    hipMemsetAsync(p, 0, n_dw * 4, 0); hipDeviceSynchronize();
    hipEventRecord(a); hipMemsetAsync(p, 1, n_dw * 4, 0); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); printf("hipMemset: %.3f ms %.2f TB/s\n", ms, n_dw * 4 / ms / 1e9);
    return 0;
}
