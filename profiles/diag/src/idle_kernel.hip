// Diagnostic: a one-wave kernel that holds its stream for `us` microseconds (s_memrealtime, 100 MHz) and does nothing.
// Used by profiles/diag/duty_probe.py to put idle time between batches: does the chip give the time back as clock?
#include <hip/hip_runtime.h>
__global__ void idle_kernel(int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int idle_us(void* stream, int us) {
    idle_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(us * 100);
    return (int)hipGetLastError();
}
