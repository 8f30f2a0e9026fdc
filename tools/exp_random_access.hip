// Kernel study (round 3): what one random access per lane costs on this part, by table size (L2 / Infinity Cache / HBM resident)
// and by access kind -- the numbers the open-address table designs in DESIGN.md are priced with.
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/exp_random_access tools/exp_random_access.hip && tools/build/exp_random_access
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

enum Op { LOAD16 = 0, LOAD8, LOAD4, STORE4, STORE16, CAS8_HIT, CAS8_MISS, MIN4, LOAD16_THEN_MIN4, NOPS };
static const char *op_name[NOPS] = {"load16", "load8", "load4", "store4", "store16", "cas8(succeeds)", "cas8(fails)", "atomicMin4", "load16+atomicMin4 same line"};

template <int OP, int U>
__global__ void __launch_bounds__(256) k(uint64_t *tab, uint64_t mask16, uint64_t n, uint64_t salt, uint64_t *sink)
{
    // mask16: number of 16-byte slots - 1
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
        uint64_t s[U];
#pragma unroll
        for (int u = 0; u < U; u++) s[u] = mix((i + u * stride) ^ salt) & mask16;
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint64_t *p = tab + 2 * s[u];
            if (OP == LOAD16) {
                const ulonglong2 v = *(const ulonglong2 *)p;
                acc += v.x ^ v.y;
            }
            else if (OP == LOAD8) acc += *p;
            else if (OP == LOAD4) acc += *(const uint32_t *)p;
            else if (OP == STORE4) *(uint32_t *)p = (uint32_t)i;
            else if (OP == STORE16) *(ulonglong2 *)p = make_ulonglong2(i, i);
            else if (OP == CAS8_HIT) acc += atomicCAS((unsigned long long *)p, 0ull, (unsigned long long)(i | 1));
            else if (OP == CAS8_MISS) acc += atomicCAS((unsigned long long *)p, ~0ull, (unsigned long long)i);
            else if (OP == MIN4) acc += atomicMin((unsigned int *)p + 2, (unsigned int)i);
            else if (OP == LOAD16_THEN_MIN4) {
                const ulonglong2 v = *(const ulonglong2 *)p;
                if (v.x != 12345) acc += atomicMin((unsigned int *)p + 2, (unsigned int)i);
            }
        }
    }
    if (acc == 0x1234567812345678ull) *sink = acc;
}

template <int OP>
static double run(uint64_t *tab, uint64_t slots, uint64_t n, uint64_t *sink, int blocks)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        if (OP == CAS8_HIT) CK(hipMemsetAsync(tab, 0, slots * 16, 0));
        CK(hipEventRecord(a, 0));
        k<OP, 4><<<blocks, 256>>>(tab, slots - 1, n, 0x1111ull * (it + 1), sink);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const uint64_t n = 1ull << 27;   // 134 M accesses per launch
    const uint64_t max_bytes = 8ull << 30;
    uint64_t *tab, *sink;
    CK(hipMalloc(&tab, max_bytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(tab, 0xff, max_bytes));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8;
    printf("device %s, %d CUs, grid %d x 256, %llu random accesses per launch, 4 independent accesses per lane in flight\n", prop.name, prop.multiProcessorCount, blocks,
           (unsigned long long)n);
    printf("%-32s", "table bytes");
    const uint64_t sizes[] = {1ull << 20, 2ull << 20, 8ull << 20, 32ull << 20, 128ull << 20, 512ull << 20, 2ull << 30, 8ull << 30};
    for (uint64_t sz : sizes) printf("%9lluM", (unsigned long long)(sz >> 20));
    printf("   (G accesses/s)\n");
    for (int op = 0; op < NOPS; op++) {
        printf("%-32s", op_name[op]);
        for (uint64_t sz : sizes) {
            const uint64_t slots = sz / 16;
            double ms = 0;
            switch (op) {
            case LOAD16: ms = run<LOAD16>(tab, slots, n, sink, blocks); break;
            case LOAD8: ms = run<LOAD8>(tab, slots, n, sink, blocks); break;
            case LOAD4: ms = run<LOAD4>(tab, slots, n, sink, blocks); break;
            case STORE4: ms = run<STORE4>(tab, slots, n, sink, blocks); break;
            case STORE16: ms = run<STORE16>(tab, slots, n, sink, blocks); break;
            case CAS8_HIT: ms = run<CAS8_HIT>(tab, slots, n, sink, blocks); break;
            case CAS8_MISS: ms = run<CAS8_MISS>(tab, slots, n, sink, blocks); break;
            case MIN4: ms = run<MIN4>(tab, slots, n, sink, blocks); break;
            case LOAD16_THEN_MIN4: ms = run<LOAD16_THEN_MIN4>(tab, slots, n, sink, blocks); break;
            }
            printf("%10.1f", (double)n / (ms * 1e-3) / 1e9);
            fflush(stdout);
        }
        printf("\n");
    }
    return 0;
}
