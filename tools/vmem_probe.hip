// Micro-benchmark of the vector memory pipe for divergent 16-byte loads:
// how many dwordx4 wave-instructions per clock a CU sustains when the 64 lanes
// hit the same address, one record each in a small (cache-resident) table, or
// one record each in a large table. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int kLoads, int kMode>
__global__ void __launch_bounds__(256) k_probe(const uint4 *__restrict__ buf, uint32_t n_records, int iters,
                                               uint32_t rec_words, uint32_t *out)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it)
    {
        s = s * 1664525u + 1013904223u;
        uint32_t r;
        if (kMode == 0) r = (uint32_t)it % n_records;                        // wave-uniform
        else if (kMode == 1) r = (s >> 8) % n_records;                       // random per lane
        else r = ((s >> 8) % (n_records / 64u)) * 64u + (threadIdx.x & 63u); // random per wave, lanes adjacent
        const uint4 *p = buf + (size_t)r * rec_words;
        uint4 v[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) acc ^= v[k].x + v[k].y + v[k].z + v[k].w;
        s ^= acc & 1u; // dependent: one round trip per iteration
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int kLoads, int kMode>
void run(const char *name, const uint4 *buf, size_t bytes, uint32_t rec_words, uint32_t *out, int blocks)
{
    const uint32_t n_records = (uint32_t)(bytes / (16 * rec_words));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_probe<kLoads, kMode><<<blocks, 256>>>(buf, n_records, 100, rec_words, out);
    hipEventRecord(e0);
    k_probe<kLoads, kMode><<<blocks, 256>>>(buf, n_records, iters, rec_words, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * kLoads;
    const double per_cu_per_s = wave_instr / 256.0 / (ms * 1e-3);
    printf("%-44s %8.3f ms  %7.2f G lane-loads/s  %6.1f clk/wave-instr/CU @2.4GHz  %7.1f GB/s useful\n", name, ms,
           wave_instr * 64 / (ms * 1e-3) / 1e9, 2.4e9 / per_cu_per_s, wave_instr * 64 * 16 / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t big = 1ull << 30;
    uint4 *buf;
    uint32_t *out;
    hipMalloc(&buf, big);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, big);
    const int blocks = 256 * 8;
    const size_t sizes[] = {16 << 10, 2 << 20, 10 << 20, 48 << 20, 1 << 30};
    for (size_t sz : sizes)
    {
        printf("-- table %zu KiB\n", sz >> 10);
        run<1, 0>("1 x16B, wave-uniform", buf, sz, 1, out, blocks);
        run<1, 2>("1 x16B, lanes adjacent", buf, sz, 1, out, blocks);
        run<1, 1>("1 x16B, random per lane", buf, sz, 1, out, blocks);
        run<5, 1>("5 x16B (80 B record), random per lane", buf, sz, 5, out, blocks);
        run<4, 1>("4 x16B (64 B record), random per lane", buf, sz, 4, out, blocks);
        run<3, 1>("3 x16B (48 B record), random per lane", buf, sz, 3, out, blocks);
        run<2, 1>("2 x16B (32 B record), random per lane", buf, sz, 2, out, blocks);
    }
    return 0;
}
