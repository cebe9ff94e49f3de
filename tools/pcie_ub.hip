// tools/pcie_ub.hip -- what host-memory choices cost on this box: hipHostRegister / hipHostMalloc / hipMalloc times and
// H2D / D2H rates from pageable, registered and hipHostMalloc'ed memory.  Build: hipcc -O2 --offload-arch=gfx950 -o scratch/pcie_ub tools/pcie_ub.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const size_t MB = 1 << 20, n = (argc > 1 ? atol(argv[1]) : 1024) * MB;
    void *d = NULL; double t;
    t = now(); CK(hipMalloc(&d, n)); printf("hipMalloc %zu MB: %.2f ms\n", n / MB, (now() - t) * 1e3);
    t = now(); CK(hipMemset(d, 0, n)); CK(hipDeviceSynchronize()); printf("first memset: %.2f ms\n", (now() - t) * 1e3);
    char *p = (char *)malloc(n); t = now(); memset(p, 1, n); printf("host first touch %zu MB: %.2f ms\n", n / MB, (now() - t) * 1e3);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int k = 0; k < 3; k++) { t = now(); CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); printf("H2D pageable: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, n / (now() - t) / 1e9); }
    for (int k = 0; k < 2; k++) { t = now(); CK(hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); printf("D2H pageable: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, n / (now() - t) / 1e9); }
    t = now(); CK(hipHostRegister(p, n, hipHostRegisterDefault)); printf("hipHostRegister %zu MB: %.2f ms\n", n / MB, (now() - t) * 1e3);
    for (int k = 0; k < 3; k++) { t = now(); CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); printf("H2D registered: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, n / (now() - t) / 1e9); }
    for (int k = 0; k < 2; k++) { t = now(); CK(hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); printf("D2H registered: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, n / (now() - t) / 1e9); }
    t = now(); CK(hipHostUnregister(p)); printf("hipHostUnregister: %.2f ms\n", (now() - t) * 1e3);
    void *h = NULL; t = now(); CK(hipHostMalloc(&h, n, hipHostMallocDefault)); printf("hipHostMalloc %zu MB: %.2f ms\n", n / MB, (now() - t) * 1e3);
    t = now(); memset(h, 2, n); printf("first touch of it: %.2f ms\n", (now() - t) * 1e3);
    t = now(); memcpy(h, p, n); printf("memcpy pageable -> pinned, 1 thread: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, n / (now() - t) / 1e9);
    for (int k = 0; k < 3; k++) { t = now(); CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); printf("H2D hipHostMalloc: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, n / (now() - t) / 1e9); }
    // small chunks on two streams (the pipelined form): 16 x 64 MB alternating
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    t = now(); for (size_t o = 0, k = 0; o < n; o += 64 * MB, k++) CK(hipMemcpyAsync((char *)d + o, (char *)h + o, (n - o < 64 * MB ? n - o : 64 * MB), hipMemcpyHostToDevice, (k & 1) ? s2 : s));
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); printf("H2D pinned, 64 MB chunks on 2 streams: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, n / (now() - t) / 1e9);
    t = now(); CK(hipHostFree(h)); printf("hipHostFree: %.2f ms\n", (now() - t) * 1e3);
    t = now(); CK(hipFree(d)); printf("hipFree: %.2f ms\n", (now() - t) * 1e3);
    t = now(); CK(hipMalloc(&d, n)); printf("hipMalloc again: %.2f ms\n", (now() - t) * 1e3); CK(hipFree(d));
    free(p);
    return 0;
}
