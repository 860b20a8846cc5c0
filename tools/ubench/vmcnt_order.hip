// Micro-test: do vector-memory loads and stores retire in issue order on gfx950 (is
// "s_waitcnt vmcnt(N)" with N younger STORES enough to guarantee an older LOAD has landed)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* cold, unsigned* hot, unsigned* out, size_t stride_words, int rounds) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned bad = 0;
    for (int r = 0; r < rounds; r++) {
        const unsigned* src = cold + ((t * 131 + (size_t)r * 7919) % (1u << 22)) * stride_words;
        unsigned v = 0xDEADBEEFu;
        unsigned* h = hot + (t & 1023) * 16;
        asm volatile(
            "global_load_dword %0, %1, off\n\t"
            "global_store_dword %2, %3, off\n\t"
            "global_store_dword %2, %3, off offset:4\n\t"
            "global_store_dword %2, %3, off offset:8\n\t"
            "global_store_dword %2, %3, off offset:12\n\t"
            "global_store_dword %2, %3, off offset:16\n\t"
            "global_store_dword %2, %3, off offset:20\n\t"
            "global_store_dword %2, %3, off offset:24\n\t"
            "global_store_dword %2, %3, off offset:28\n\t"
            "s_waitcnt vmcnt(8)\n\t"
            : "+v"(v) : "v"(src), "v"(h), "v"((unsigned)r) : "memory");
        const unsigned expect = (unsigned)((src - cold) * 2654435761u);
        if (v != expect) bad++;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    out[t] = bad;
}
__global__ void fill(unsigned* cold, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        cold[i] = (unsigned)(i * 2654435761u);
}
int main() {
    const size_t stride_words = 64;            // 256 B apart
    const size_t n = (size_t)(1u << 22) * stride_words;   // 1 GiB of words
    unsigned *cold, *hot, *out;
    hipMalloc(&cold, n * 4); hipMalloc(&hot, 1024 * 16 * 4 + 64); 
    const int blocks = 2048, threads = 256;
    hipMalloc(&out, blocks * threads * 4);
    fill<<<4096, 256>>>(cold, n);
    probe<<<blocks, threads>>>(cold, hot, out, stride_words, 64);
    std::vector<unsigned> h(blocks * threads);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long bad = 0; for (auto x : h) bad += x;
    printf("loads checked %llu, stale after vmcnt(#younger stores): %llu\n", (unsigned long long)h.size() * 64, bad);
    return 0;
}
