// Store microbenchmark (MI355X): ONE WAVE PER REGION, each wave writing its region row after row (the shape of
// mrp_cross_emit_kernel: a wave owns a column's cells and writes two arrays, 8 B per lane and row) against the same bytes
// written by a grid-stride loop.  build: hipcc -O3 --offload-arch=gfx950 -o store_streams store_streams.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed, aligned(4))) u2 { uint32_t x, y; };
// mode 0: a wave per region, rows of `lanes` x 8 B, two arrays; mode 1: same, one array; mode 2: same as 0 but regions start 128-byte aligned
__global__ void __launch_bounds__(64) per_wave(uint32_t *np, uint32_t *cost, int64_t region_dw, int rows, int lanes, int arrays, int delay, int use_lds) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1600]; // 6.4 KB a wave, as the kernel's tables
    const int lane = threadIdx.x;
    const int64_t base = (int64_t) blockIdx.x * region_dw;
    uint32_t acc = lane;
    if (use_lds) { for (int i = lane; i < 1600; i += 64) lds[i] = i * 2654435761u; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    for (int r = 0; r < rows; r++) {
        if (use_lds) { // two wave-uniform 16-byte LDS reads a row, through scalar registers (the table row and the transition terms)
            const uint4 aq = *reinterpret_cast<const uint4 *>(lds + 8 * (r & 63)), tq = *reinterpret_cast<const uint4 *>(lds + 1024 + 4 * (r & 63));
            acc += (uint32_t) __builtin_amdgcn_readfirstlane((int) aq.x) + (uint32_t) __builtin_amdgcn_readfirstlane((int) aq.y) +
                   ((int) __builtin_amdgcn_readfirstlane((int) tq.x) < 0 ? lane : 2 * lane) + (uint32_t) __builtin_amdgcn_readfirstlane((int) tq.z);
        }
        for (int d = 0; d < delay; d++) acc = acc * 1664525u + 1013904223u; // stand-in for the cell's arithmetic
        if (lane < lanes) {
            const int64_t o = base + 2 * ((int64_t) r * lanes + lane);
            u2 v = {acc, (uint32_t) r};
            *(u2 *) (np + o) = v;
            if (arrays > 1) *(u2 *) (cost + o) = v;
        }
    }
}
int main() {
    const int64_t n_dw = (int64_t) 1 << 29; // 2 GiB per array
    uint32_t *a, *b; hipMalloc(&a, n_dw * 4 + 4096); hipMalloc(&b, n_dw * 4 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, int lanes, int rows, int arrays, int delay, int align_dw, int use_lds) {
        int64_t region = 2ll * lanes * rows; // dwords
        if (align_dw) region = (region + align_dw - 1) / align_dw * align_dw; else region += 1; // odd: rows start anywhere
        const int64_t waves = n_dw / region;
        per_wave<<<(unsigned) waves, 64>>>(a + (align_dw ? 0 : 1), b + (align_dw ? 0 : 1), region, rows, lanes, arrays, delay, use_lds); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) per_wave<<<(unsigned) waves, 64>>>(a + (align_dw ? 0 : 1), b + (align_dw ? 0 : 1), region, rows, lanes, arrays, delay, use_lds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        const double bytes = (double) waves * 2.0 * lanes * rows * 4.0 * arrays;
        printf("%-34s lanes %2d rows %3d arrays %d delay %3d: %7.3f ms  %5.2f TB/s  (%lld waves)\n", name, lanes, rows, arrays, delay, ms, bytes / ms / 1e9, (long long) waves);
    };
    for (int delay : {0, 8, 32}) {
        run("wave per region, unaligned", 50, 25, 2, delay, 0, 0);
        run("wave per region, 128 B aligned", 50, 25, 2, delay, 32, 0);
        run("wave per region, one array", 50, 25, 1, delay, 0, 0);
        run("wave per region, 64 lanes", 64, 25, 2, delay, 0, 0);
        run("wave per region, 100 x 50 rows", 50, 50, 2, delay, 0, 0);
        run("wave per region, short (6 rows)", 50, 6, 2, delay, 0, 0);
        run("... with uniform LDS reads", 50, 25, 2, delay, 0, 1);
        run("... with uniform LDS reads, long", 50, 50, 2, delay, 0, 1);
    }
    return 0;
}
