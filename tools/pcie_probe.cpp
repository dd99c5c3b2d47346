// pcie_probe: what a host-pointer entry point can expect from the host <-> device link on this box.
// Measures, for 32 MB and 320 MB buffers: pageable H2D / D2H (touched and untouched destination pages), pinned
// (hipHostMalloc) H2D / D2H, hipHostRegister of caller memory + DMA, CPU memcpy into pinned staging with 1 / 2 / 4 threads,
// and the first-touch cost of fresh pages.  Build: hipcc -O2 -o gpurun_out/pcie_probe tools/pcie_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

static void *fresh(size_t bytes) {   // untouched anonymous pages, like a fresh np.empty / Vec::with_capacity
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(1); }
    return p;
}
static void par_memcpy(uint8_t *dst, const uint8_t *src, size_t bytes, int threads) {
    if (threads <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    size_t per = (bytes / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; t++) {
        size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo < hi) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto &t : th) t.join();
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t mb : {32, 320}) {
        const size_t bytes = mb << 20;
        void *d; CK(hipMalloc(&d, bytes));
        void *pin; CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
        memset(pin, 1, bytes);
        uint8_t *page = (uint8_t *)fresh(bytes);
        double t0 = now(); memset(page, 2, bytes); double t_touch = now() - t0;
        printf("---- %zu MB ----\n", mb);
        printf("first touch (memset of fresh pages)      %7.2f ms  %6.1f GB/s\n", t_touch * 1e3, bytes / t_touch / 1e9);
        auto timeit = [&](const char *what, auto fn, int reps = 3) {
            double best = 1e9;
            for (int r = 0; r < reps; r++) { double a = now(); fn(); double b = now() - a; if (b < best) best = b; }
            printf("%-40s %7.2f ms  %6.1f GB/s\n", what, best * 1e3, bytes / best / 1e9);
        };
        timeit("H2D pinned", [&] { CK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
        timeit("D2H pinned", [&] { CK(hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
        timeit("H2D pageable (touched)", [&] { CK(hipMemcpyAsync(d, page, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
        timeit("D2H pageable (touched)", [&] { CK(hipMemcpyAsync(page, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
        {
            double best = 1e9;
            for (int r = 0; r < 3; r++) {
                uint8_t *f = (uint8_t *)fresh(bytes);
                double a = now(); CK(hipMemcpyAsync(f, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double b = now() - a;
                if (b < best) best = b;
                munmap(f, bytes);
            }
            printf("%-40s %7.2f ms  %6.1f GB/s\n", "D2H pageable (UNTOUCHED destination)", best * 1e3, bytes / best / 1e9);
        }
        {
            double a = now(); CK(hipHostRegister(page, bytes, hipHostRegisterDefault)); double b = now() - a;
            printf("%-40s %7.2f ms  %6.1f GB/s\n", "hipHostRegister (touched pages)", b * 1e3, bytes / b / 1e9);
            timeit("H2D registered", [&] { CK(hipMemcpyAsync(d, page, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
            timeit("D2H registered", [&] { CK(hipMemcpyAsync(page, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
            a = now(); CK(hipHostUnregister(page)); b = now() - a;
            printf("%-40s %7.2f ms\n", "hipHostUnregister", b * 1e3);
        }
        for (int th : {1, 2, 4, 8}) {
            char what[64]; snprintf(what, sizeof what, "memcpy pageable -> pinned, %d thread(s)", th);
            timeit(what, [&] { par_memcpy((uint8_t *)pin, page, bytes, th); });
        }
        for (int th : {1, 2, 4}) {
            char what[64]; snprintf(what, sizeof what, "memcpy pinned -> pageable, %d thread(s)", th);
            timeit(what, [&] { par_memcpy(page, (uint8_t *)pin, bytes, th); });
        }
        {   // chunked pipeline: CPU copies chunk i+1 into a pinned ring while chunk i is DMA'd
            const size_t CH = 4 << 20; const int NB = 4;
            uint8_t *ring; CK(hipHostMalloc((void **)&ring, CH * NB, hipHostMallocDefault));
            hipEvent_t ev[NB]; for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            timeit("H2D pageable via 4 x 4 MB pinned ring, 1 thread", [&] {
                size_t nchunk = bytes / CH;
                for (size_t c = 0; c < nchunk; c++) {
                    int s = (int)(c % NB);
                    if (c >= (size_t)NB) CK(hipEventSynchronize(ev[s]));
                    memcpy(ring + s * CH, page + c * CH, CH);
                    CK(hipMemcpyAsync((uint8_t *)d + c * CH, ring + s * CH, CH, hipMemcpyHostToDevice, st));
                    CK(hipEventRecord(ev[s], st));
                }
                CK(hipStreamSynchronize(st));
            });
            timeit("D2H pageable via 4 x 4 MB pinned ring, 1 thread", [&] {
                size_t nchunk = bytes / CH;
                for (size_t c = 0; c < nchunk + NB; c++) {
                    if (c >= (size_t)NB) { size_t k = c - NB; int s = (int)(k % NB); CK(hipEventSynchronize(ev[s])); memcpy(page + k * CH, ring + s * CH, CH); }
                    if (c < nchunk) { int s = (int)(c % NB); CK(hipMemcpyAsync(ring + s * CH, (uint8_t *)d + c * CH, CH, hipMemcpyDeviceToHost, st)); CK(hipEventRecord(ev[s], st)); }
                }
            });
            hipHostFree(ring);
        }
        // both directions at once on two streams (pinned): does the link run full duplex?
        {
            hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            void *d2, *pin2; CK(hipMalloc(&d2, bytes)); CK(hipHostMalloc(&pin2, bytes, hipHostMallocDefault));
            timeit("H2D + D2H concurrently (pinned; per direction)", [&] {
                CK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(pin2, d2, bytes, hipMemcpyDeviceToHost, s2));
                CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(s2));
            });
            hipFree(d2); hipHostFree(pin2); hipStreamDestroy(s2);
        }
        munmap(page, bytes); hipHostFree(pin); hipFree(d);
    }
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    return 0;
}
