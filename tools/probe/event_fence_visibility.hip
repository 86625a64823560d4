// Are a kernel's plain stores visible to a kernel of ANOTHER stream that waits for an event created with hipEventDisableSystemFence?
// Stream A writes value i into 64 MB (every XCD's L2 ends up holding dirty lines), records the event; stream B waits for it and
// checks every word, then writes i back into a second buffer that A's next iteration checks: both directions, thousands of rounds.
//   hipcc --offload-arch=gfx950 -O3 -w tools/probe/event_fence_visibility.hip -o tools/probe/event_fence_visibility
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void fill(unsigned* p, size_t n, unsigned v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i;
}
__global__ void check(const unsigned* p, size_t n, unsigned v, unsigned long long* bad)
{
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != v + (unsigned)i;
    if (b) atomicAdd(bad, b);
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 3000;
    const size_t n = 16u << 20;     // 64 MB
    for (unsigned flags : {(unsigned)hipEventDisableTiming, (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence)}) {
        hipStream_t A, B;
        hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
        hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
        unsigned *x, *y;
        unsigned long long* bad;
        hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&bad, 8);
        hipMemset(bad, 0, 8); hipMemset(y, 0, n * 4);
        hipEvent_t ea, eb;
        hipEventCreateWithFlags(&ea, flags);
        hipEventCreateWithFlags(&eb, flags);
        hipDeviceSynchronize();
        for (int r = 1; r <= rounds; r++) {
            fill<<<2048, 256, 0, A>>>(x, n, (unsigned)r * 2654435761u);
            hipEventRecord(ea, A);
            hipStreamWaitEvent(B, ea, 0);
            check<<<2048, 256, 0, B>>>(x, n, (unsigned)r * 2654435761u, bad);
            fill<<<2048, 256, 0, B>>>(y, n, (unsigned)r * 40503u);
            hipEventRecord(eb, B);
            hipStreamWaitEvent(A, eb, 0);
            check<<<2048, 256, 0, A>>>(y, n, (unsigned)r * 40503u, bad);
        }
        hipDeviceSynchronize();
        unsigned long long h = 0;
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("event flags 0x%08x: %d rounds x 2 directions x 64 MB, wrong words: %llu\n", flags, rounds, h);
        hipFree(x); hipFree(y); hipFree(bad);
    }
    return 0;
}
