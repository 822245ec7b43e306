// Vector-memory issue cost per wave instruction by load width and by number of
// active lanes, all hitting a 16 KiB (L1-resident) table.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int kWords, int kMode>
__global__ void __launch_bounds__(256) k_probe(const uint32_t *__restrict__ buf, uint32_t n_records, int iters,
                                               uint32_t active, uint32_t *out)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    if ((threadIdx.x & 63u) < active)
    {
        for (int it = 0; it < iters; ++it)
        {
            s = s * 1664525u + 1013904223u;
            const uint32_t r = kMode == 0 ? (uint32_t)it % n_records : (s >> 8) % n_records;
            const uint32_t *p = buf + (size_t)r * 4u;
            if (kWords == 1) { acc ^= *p; }
            if (kWords == 2) { const uint2 v = *(const uint2 *)p; acc ^= v.x + v.y; }
            if (kWords == 3) { typedef uint32_t u3 __attribute__((ext_vector_type(3))); const u3 v = *(const u3 *)p; acc ^= v.x + v.y + v.z; }
            if (kWords == 4) { const uint4 v = *(const uint4 *)p; acc ^= v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int kWords, int kMode>
void run(const char *name, const uint32_t *buf, uint32_t active, uint32_t *out)
{
    const int blocks = 256 * 8, iters = 4000;
    const uint32_t n_records = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_probe<kWords, kMode><<<blocks, 256>>>(buf, n_records, 100, active, out);
    hipEventRecord(e0);
    k_probe<kWords, kMode><<<blocks, 256>>>(buf, n_records, iters, active, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters;
    printf("%-28s active %2u  %7.3f ms  %6.1f clk/wave-instr/CU @2.4GHz\n", name, active, ms,
           2.4e9 / (wave_instr / 256.0 / (ms * 1e-3)));
}

int main()
{
    uint32_t *buf, *out;
    hipMalloc(&buf, 1 << 20);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, 1 << 20);
    run<1, 0>("dword uniform", buf, 64, out);
    run<2, 0>("dwordx2 uniform", buf, 64, out);
    run<3, 0>("dwordx3 uniform", buf, 64, out);
    run<4, 0>("dwordx4 uniform", buf, 64, out);
    run<1, 1>("dword random", buf, 64, out);
    run<2, 1>("dwordx2 random", buf, 64, out);
    run<3, 1>("dwordx3 random", buf, 64, out);
    run<4, 1>("dwordx4 random", buf, 64, out);
    for (uint32_t a : {48u, 32u, 16u, 8u, 1u})
    {
        run<4, 1>("dwordx4 random", buf, a, out);
        run<4, 0>("dwordx4 uniform", buf, a, out);
        run<1, 1>("dword random", buf, a, out);
    }
    return 0;
}
