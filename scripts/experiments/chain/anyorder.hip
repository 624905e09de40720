// Does hipExtAnyOrderLaunch drop the barrier bit on gfx950?  Kernel A spins ~200 us, kernel B stamps its start.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long* out, long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
__global__ void stamp(long long* out) { if (threadIdx.x == 0) out[2] = wall_clock64(); }
int main()
{
    long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int flags = 0; flags <= 1; flags++) {
        for (int rep = 0; rep < 3; rep++) {
            hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, d, 20000LL); // 100 MHz clock: 200 us
            hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, d);
            hipStreamSynchronize(s);
            long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("flags %d: A ran %lld ticks; B started %lld ticks after A started (%s)\n", flags, h[1] - h[0], h[2] - h[0], h[2] < h[1] ? "OVERLAP" : "after A");
        }
    }
    return 0;
}
