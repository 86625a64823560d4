// What does a cross-stream ordering point cost the stream that carries it?  Two kernels of ~20 us follow each other on stream A a few
// hundred times; between them: nothing / hipStreamWaitEvent on an event the other stream recorded long ago / hipEventRecord /
// hipStreamWaitValue32 on a word the other stream wrote (hipStreamWriteValue32) / both directions.
//   hipcc --offload-arch=gfx950 -O3 -w tools/probe/marker_cost.hip -o tools/probe/marker_cost && tools/probe/marker_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>

__global__ void spin(float* p, int n)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < n; i++) v = __builtin_fmaf(v, 0.999f, 0.5f);
    p[threadIdx.x] = v;
}
__global__ void tiny(float* p) { p[0] += 1.f; }

int main()
{
    hipStream_t A, B;
    hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    float *pa, *pb;
    hipMalloc(&pa, 1 << 20); hipMalloc(&pb, 1 << 20);
    hipMemset(pa, 0, 1 << 20); hipMemset(pb, 0, 1 << 20);
    uint32_t* word = nullptr;
    const bool have_word = hipExtMallocWithFlags((void**)&word, 64, hipMallocSignalMemory) == hipSuccess;
    if (have_word) hipMemset(word, 0, 64);
    const int iters = 300, n = 1200;
    std::vector<hipEvent_t> evs(2 * iters);
  for (unsigned flags : {(unsigned)hipEventDisableTiming, (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence),
                         (unsigned)(hipEventDisableTiming | hipEventReleaseToDevice)}) {
    printf("event flags 0x%x\n", flags);
    for (auto& e : evs) if (hipEventCreateWithFlags(&e, flags) != hipSuccess) { printf("  cannot create\n"); return 1; }
    const char* names[] = {"nothing between the two kernels", "hipStreamWaitEvent (event of the other stream, recorded a step earlier)",
                           "hipEventRecord (for the other stream to wait on)", "record + wait (both directions)",
                           "hipStreamWaitValue32 (word the other stream wrote a step earlier)", "hipStreamWriteValue32 + hipStreamWaitValue32"};
    for (int mode = 0; mode < 6; mode++) {
        if (mode >= 4 && !have_word) { printf("%-80s  (no signal memory)\n", names[mode]); continue; }
        if (have_word) hipMemset(word, 0, 64);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; i++) {
                // the other stream: a little work, then its signal for the NEXT iteration of A (so A never actually has to wait)
                tiny<<<1, 64, 0, B>>>(pb);
                if (mode == 1 || mode == 3) hipEventRecord(evs[2 * i], B);
                if (mode >= 4) hipStreamWriteValue32(B, word, (uint32_t)(i + 1), 0);
                spin<<<256, 256, 0, A>>>(pa, n);
                if (mode == 1 || mode == 3) { if (i > 0) hipStreamWaitEvent(A, evs[2 * (i - 1)], 0); }
                if (mode == 2 || mode == 3) { hipEventRecord(evs[2 * i + 1], A); hipStreamWaitEvent(B, evs[2 * i + 1], 0); }
                if (mode == 4 || mode == 5) { if (i > 0) hipStreamWaitValue32(A, word, (uint32_t)i, hipStreamWaitValueGte, 0xFFFFFFFFu); }
                spin<<<256, 256, 0, A>>>(pa, n);
            }
            hipStreamSynchronize(A);
            hipStreamSynchronize(B);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 1) printf("%-80s %8.2f us per iteration (two kernels)\n", names[mode], us / iters);
        }
    }
    for (auto& e : evs) hipEventDestroy(e);
  }
    return 0;
}
