// Issue cost of buffer_load_dwordx4 (descriptor + 32-bit offset) against
// global_load_dwordx4 (64-bit address per lane), 16 KiB L1-resident table.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int kKind, int kMode>
__global__ void __launch_bounds__(256) k_probe(const uint4 *__restrict__ buf, uint32_t n_records, int iters,
                                               uint32_t active, uint32_t *out)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, n_records * 16u, 0x00020000);
    if ((threadIdx.x & 63u) < active)
    {
        for (int it = 0; it < iters; ++it)
        {
            s = s * 1664525u + 1013904223u;
            const uint32_t r = kMode == 0 ? (uint32_t)it % n_records : (s >> 8) % n_records;
            if (kKind == 0)
            {
                const uint4 v = buf[r];
                acc ^= v.x + v.y + v.z + v.w;
            }
            else
            {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, r * 16u, 0, 0);
                acc ^= v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int kKind, int kMode>
void run(const char *name, const uint4 *buf, uint32_t active, uint32_t *out)
{
    const int blocks = 256 * 8, iters = 4000;
    const uint32_t n_records = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_probe<kKind, kMode><<<blocks, 256>>>(buf, n_records, 100, active, out);
    hipEventRecord(e0);
    k_probe<kKind, kMode><<<blocks, 256>>>(buf, n_records, iters, active, out);
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
    uint4 *buf;
    uint32_t *out;
    hipMalloc(&buf, 1 << 20);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, 1 << 20);
    for (uint32_t a : {64u, 16u})
    {
        run<0, 0>("global x4 uniform", buf, a, out);
        run<1, 0>("buffer x4 uniform", buf, a, out);
        run<0, 1>("global x4 random", buf, a, out);
        run<1, 1>("buffer x4 random", buf, a, out);
    }
    return 0;
}
