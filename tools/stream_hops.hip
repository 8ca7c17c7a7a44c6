// Dev tool: what a cross-stream dependency costs on this GPU. A chain of N tiny kernels, (a) all on one stream, (b) alternating
// between two streams with hipEventRecord / hipStreamWaitEvent between consecutive kernels, (c) the same over three streams;
// wall time of the chain / N = per-link cost. The strip frame's critical chain crosses streams ~11 times per frame.
//   hipcc --offload-arch=gfx950 -O2 -o tools/stream_hops tools/stream_hops.hip && GPU_MAX_HW_QUEUES=8 tools/stream_hops
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void busy(int* p, int n) { int s = 0; for (int i = 0; i < n; ++i) s += __builtin_amdgcn_readfirstlane(i) ^ s; if (s == 123456789) p[1] = s; }
static double run(int nstreams, int links, int* d, int busy_iters)
{
    std::vector<hipStream_t> st(nstreams);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(links);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep)
    {
        for (auto& s : st) hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < links; ++i)
        {
            hipStream_t s = st[i % nstreams];
            if (i > 0 && nstreams > 1) hipStreamWaitEvent(s, ev[i - 1], 0);
            if (busy_iters) busy<<<256, 256, 0, s>>>(d, busy_iters); else tiny<<<1, 64, 0, s>>>(d);
            if (nstreams > 1) hipEventRecord(ev[i], s);
        }
        for (auto& s : st) hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    for (auto& s : st) hipStreamDestroy(s);
    for (auto& e : ev) hipEventDestroy(e);
    return best / links;
}
// one stream, `links` tiny kernels, and between consecutive kernels: `recs` hipEventRecord on that stream and `waits`
// hipStreamWaitEvent on an event of ANOTHER stream that completed long ago: what bookkeeping between two kernels costs
static double run_marks(int links, int recs, int waits, int* d)
{
    hipStream_t s, other;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&other, hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(8);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t old_ev;
    hipEventCreateWithFlags(&old_ev, hipEventDisableTiming);
    tiny<<<1, 64, 0, other>>>(d);
    hipEventRecord(old_ev, other);
    hipStreamSynchronize(other);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep)
    {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < links; ++i)
        {
            for (int k = 0; k < waits; ++k) hipStreamWaitEvent(s, old_ev, 0);
            for (int k = 0; k < recs; ++k) hipEventRecord(ev[k & 7], s);
            tiny<<<1, 64, 0, s>>>(d);
        }
        hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    hipStreamDestroy(s); hipStreamDestroy(other);
    for (auto& e : ev) hipEventDestroy(e);
    hipEventDestroy(old_ev);
    return best / links;
}
int main()
{
    int* d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    const int links = 400;
    printf("one stream, per kernel: plain %.2f us; + 1 event record %.2f; + 4 records %.2f; + 1 wait on a long-completed event of another stream %.2f; + 4 waits %.2f; + 4 records + 3 waits %.2f\n",
           run_marks(links, 0, 0, d), run_marks(links, 1, 0, d), run_marks(links, 4, 0, d), run_marks(links, 0, 1, d), run_marks(links, 0, 4, d), run_marks(links, 4, 3, d));
    for (int busy_iters : {0, 20000})
    {
        printf("%s kernels, %d links: one stream %.2f us / link, two streams %.2f, three streams %.2f\n", busy_iters ? "~20 us" : "empty", links,
               run(1, links, d, busy_iters), run(2, links, d, busy_iters), run(3, links, d, busy_iters));
    }
    return 0;
}
