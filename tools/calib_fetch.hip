// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against a KNOWN byte count, in the access pattern of the
// row-reduction kernels: 16-byte-per-lane non-temporal streaming reads of a buffer far larger than the Infinity Cache, and
// 64-bit no-return atomic adds (the flushes into the row-sum table).  usage: calib_fetch [MiB]   (run under rocprofv3 --pmc)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef long long ll2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void calib_read(const ll2 *__restrict__ p, size_t n16, long long *out)
{
    long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const ll2 v = __builtin_nontemporal_load(p + i);
        acc += v.x + v.y;
    }
    if (acc == 0x7fffffffffffffffll) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_atomic(unsigned long long *p, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256)
        __hip_atomic_fetch_add(p + i, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;
    const size_t bytes = mib << 20;
    void *buf; long long *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int it = 0; it < 5; ++it) calib_read<<<4096, 256>>>((const ll2 *)buf, bytes / 16, out);
    const size_t abytes = 32u << 20;   // 32 MiB of 8-byte atomics, as one launch of the sweep writes
    for (int it = 0; it < 5; ++it) calib_atomic<<<2048, 256>>>((unsigned long long *)buf, abytes / 8);
    hipDeviceSynchronize();
    printf("calib_read: %zu bytes per launch; calib_atomic: %zu bytes of 8-byte atomic adds per launch\n", bytes, abytes);
    return 0;
}
