// Gathering 80-byte records (five 16-byte pieces) for `active` lanes of a wave
// from a 10 MB table, three ways:
//   A  every lane loads its own five pieces (five wave instructions whose lanes
//      all touch different cache lines);
//   B  the wave's pieces are dealt to consecutive lanes (lane s loads piece s%5
//      of record s/5), staged through LDS, and each owner reads its record back;
//   C  as B with loads that write LDS directly.
// Reports cycles per gathered wave-record-set per CU with 16 waves resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_void_ptr;

constexpr int kWaves = 4;        // per block
constexpr int kSlots = 5 * 64;   // pieces per wave

template <int kKind>
__global__ void __launch_bounds__(256, 4) k_probe(const u32x4 *__restrict__ tab, uint32_t n_records, int iters,
                                                  uint32_t active, uint32_t *out, uint32_t stride)
{
    __shared__ u32x4 s_data[kWaves][kSlots];
    __shared__ uint32_t s_idx[kWaves][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const uint32_t n_pieces = 5u * active;
    for (int it = 0; it < iters; ++it)
    {
        s = s * 1664525u + 1013904223u;
        const uint32_t r = (s >> 8) % n_records;
        if (kKind == 0)
        {
            if (lane < active)
            {
                u32x4 v[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) v[k] = tab[r * stride + k];
#pragma unroll
                for (int k = 0; k < 5; ++k) acc ^= v[k].x + v[k].y + v[k].z + v[k].w;
            }
        }
        else
        {
            if (lane < active) s_idx[wave][lane] = r;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (kKind == 1)
            {
                u32x4 v[5];
#pragma unroll
                for (int k = 0; k < 5; ++k)
                {
                    const uint32_t slot = lane + 64u * k;
                    if (64u * k < n_pieces) // wave-uniform
                    {
                        const uint32_t owner = slot / 5u, part = slot - owner * 5u;
                        const uint32_t rec = s_idx[wave][owner < 64u ? owner : 0u];
                        if (slot < n_pieces) v[k] = tab[rec * stride + part];
                    }
                }
#pragma unroll
                for (int k = 0; k < 5; ++k)
                {
                    const uint32_t slot = lane + 64u * k;
                    if (slot < n_pieces) s_data[wave][slot] = v[k];
                }
            }
            else
            {
#pragma unroll
                for (int k = 0; k < 5; ++k)
                {
                    const uint32_t slot = lane + 64u * k;
                    if (64u * k < n_pieces)
                    {
                        const uint32_t owner = slot / 5u, part = slot - owner * 5u;
                        const uint32_t rec = s_idx[wave][owner < 64u ? owner : 0u];
                        if (slot < n_pieces)
                        {
                            __builtin_amdgcn_global_load_lds((const void *)(tab + rec * stride + part),
                                                             (lds_void_ptr)&s_data[wave][64 * k], 16, 0, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < active)
            {
#pragma unroll
                for (int k = 0; k < 5; ++k)
                {
                    const u32x4 v = s_data[wave][lane * 5u + k];
                    acc ^= v.x + v.y + v.z + v.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int kKind>
uint32_t run(const char *name, const u32x4 *tab, uint32_t n_records, uint32_t active, uint32_t *out, uint32_t *h_out,
             uint32_t stride = 5)
{
    const int blocks = 256 * 4, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_probe<kKind><<<blocks, 256>>>(tab, n_records, 50, active, out, stride);
    hipEventRecord(e0);
    k_probe<kKind><<<blocks, 256>>>(tab, n_records, iters, active, out, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h_out, out, blocks * 256 * 4, hipMemcpyDeviceToHost);
    uint32_t sum = 0;
    for (int i = 0; i < blocks * 256; ++i) sum += ((uint32_t)i % 64u < active) ? h_out[i] : 0u;
    const double sets = (double)blocks * 4 * iters;
    printf("%-22s %7u records of %3u B  active %2u  %7.3f ms  %6.1f clk/wave-gather/CU @2.4GHz  %6.2f TB/s  check %08x\n", name, n_records, stride * 16u, active, ms,
           2.4e9 / (sets / 256.0 / (ms * 1e-3)), sets * active * 80.0 / (ms * 1e-3) / 1e12, sum);
    return sum;
}

int main()
{
    const uint32_t max_records = 1u << 20;
    u32x4 *tab;
    uint32_t *out;
    const size_t bytes = (size_t)max_records * 128;
    hipMalloc(&tab, bytes);
    hipMalloc(&out, 256 * 4 * 256 * 4);
    uint32_t *h = (uint32_t *)malloc(bytes);
    for (size_t i = 0; i < bytes / 4; ++i) h[i] = (uint32_t)i * 2246822519u + 7u;
    hipMemcpy(tab, h, bytes, hipMemcpyHostToDevice);
    uint32_t *h_out = (uint32_t *)malloc(256 * 4 * 256 * 4);
    for (uint32_t a : {64u, 37u, 20u, 12u})
    {
        const uint32_t c0 = run<0>("A per-lane", tab, 130603, a, out, h_out);
        const uint32_t c1 = run<1>("B dealt, via registers", tab, 130603, a, out, h_out);
        const uint32_t c2 = run<2>("C dealt, direct to LDS", tab, 130603, a, out, h_out);
        if (c0 != c1 || c0 != c2) printf("   MISMATCH\n");
    }
    // the three forms again on tables that fit L2
    for (uint32_t n : {8192u, 32768u})
    {
        run<0>("A per-lane", tab, n, 37, out, h_out);
        run<1>("B dealt, via registers", tab, n, 37, out, h_out);
        run<2>("C dealt, direct to LDS", tab, n, 37, out, h_out);
    }
    // table size and record alignment (form A, 37 lanes): 80-byte records packed, or one per 128-byte line
    for (uint32_t n : {8192u, 32768u, 130603u, 524288u, 1048576u})
    {
        run<0>("A packed", tab, n, 37, out, h_out, 5);
        run<0>("A line-aligned", tab, n, 37, out, h_out, 8);
    }
    return 0;
}
