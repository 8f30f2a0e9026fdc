// Is hipMemcpyAsync from PAGEABLE host memory done with its source when it returns?  (tgpu.h's ownership rule -- the caller may reuse its
// arrays once add_input has returned -- and every small upload of the library rely on it.)  A long kernel keeps the stream busy, the copy is
// enqueued behind it, the source is overwritten right after the call; the device must hold the ORIGINAL pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__global__ void spin(unsigned long long cycles, int *sink)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink) *sink = 1;
}
int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int bad = 0;
    for (size_t bytes : {64ul, 4096ul, 16384ul, 30000ul, 65536ul, 262144ul, 1048576ul, 8388608ul, 67108864ul}) {
        unsigned char *src = (unsigned char *)malloc(bytes), *dev, *back = (unsigned char *)malloc(bytes);
        CK(hipMalloc(&dev, bytes));
        memset(src, 0xA5, bytes);
        spin<<<1, 1, 0, s>>>(2000000ull, nullptr);   // ~20 ms at 100 MHz
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s));
        const double call_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        memset(src, 0x3C, bytes);                     // the caller reuses its array
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(back, dev, bytes, hipMemcpyDeviceToHost));
        size_t wrong = 0;
        for (size_t i = 0; i < bytes; i++) wrong += back[i] != 0xA5;
        printf("bytes %10zu: %s (%zu bytes saw the overwritten source); the call took %.0f us behind a ~20 ms kernel\n", bytes,
               wrong ? "SOURCE READ LATER" : "source consumed at return", wrong, call_us);
        bad += wrong != 0;
        CK(hipFree(dev));
        free(src);
        free(back);
    }
    return bad ? 1 : 0;
}
