// Can the CPU write straight into hipMalloc'ed device memory on this box (large BAR), and how fast?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csetjmp>
#include <csignal>
#include <unistd.h>
static sigjmp_buf g_jb;
static void on_segv(int) { siglongjmp(g_jb, 1); }
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using Clock = std::chrono::steady_clock;
static double us(Clock::time_point a) { return std::chrono::duration<double, std::micro>(Clock::now() - a).count(); }
int main() {
  const size_t bytes = 64 << 20;
  uint8_t* d = nullptr;
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 0) CK(hipMalloc(&d, bytes));
    if (mode == 1) { if (hipExtMallocWithFlags((void**)&d, bytes, hipDeviceMallocFinegrained) != hipSuccess) { printf("finegrained alloc failed\n"); continue; } }
    if (mode == 2) { if (hipExtMallocWithFlags((void**)&d, bytes, hipDeviceMallocUncached) != hipSuccess) { printf("uncached alloc failed\n"); continue; } }
    const char* nm[] = {"hipMalloc", "hipExtMallocWithFlags(Finegrained)", "hipExtMallocWithFlags(Uncached)"};
    CK(hipMemset(d, 0, bytes)); CK(hipDeviceSynchronize());
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    if (sigsetjmp(g_jb, 1) == 0) { volatile uint8_t* p = d; p[0] = 42; p[4096] = 43; if (p[0] != 42) printf("readback mismatch\n"); }
    else { printf("%-36s CPU access: NO (fault)\n", nm[mode]); signal(SIGSEGV, SIG_DFL); CK(hipFree(d)); continue; }
    signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
    // parent: write pattern with memcpy, verify via D2H copy
    std::vector<uint8_t> src(bytes), back(bytes);
    for (size_t i = 0; i < bytes; ++i) src[i] = (uint8_t)(i * 7 + 1);
    auto t0 = Clock::now();
    memcpy(d, src.data(), bytes);
    double t_all = us(t0);
    CK(hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost));
    bool ok = memcmp(back.data(), src.data(), bytes) == 0;
    // small-chunk rate: 650 KB chunks
    t0 = Clock::now();
    for (int r = 0; r < 50; ++r) memcpy(d + (size_t)r * 655360, src.data() + (size_t)r * 655360, 655360);
    double t_chunk = us(t0) / 50;
    // scattered 260-byte rows
    t0 = Clock::now();
    for (int r = 0; r < 10000; ++r) memcpy(d + (size_t)r * 260, src.data() + (size_t)(r * 37 % 10000) * 388, 260);
    double t_rows = us(t0);
    t0 = Clock::now();
    volatile uint32_t sink = 0; for (int r = 0; r < 1000; ++r) sink += ((volatile uint32_t*)d)[r * 1024];
    double t_read = us(t0) / 1000;
    printf("%-36s CPU access: YES verify=%d  64MB memcpy %.1f GB/s | 650KB chunk %.1f us (%.1f GB/s) | 10K x 260B rows %.1f us | 4B read %.2f us\n",
           nm[mode], ok, bytes / t_all / 1e3, t_chunk, 655360 / t_chunk / 1e3, t_rows, t_read);
    CK(hipFree(d));
  }
  return 0;
}
