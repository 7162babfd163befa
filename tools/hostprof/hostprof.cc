// LD_PRELOAD sampling profiler for the self-play host (no perf / gdb in the image): ITIMER_PROF at 1 kHz, the handler
// records the interrupted instruction pointer; at exit the samples that fall into libp3host.so are written as offsets
// (resolve with addr2line -f -C -e libp3host.so).  Build: g++ -O2 -shared -fPIC -o hostprof.so hostprof.cc
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <atomic>
static constexpr int kMax = 1 << 20;
static unsigned long g_ip[kMax];
static std::atomic<int> g_n{0};
static void on_prof(int, siginfo_t*, void* uc) {
  const int i = g_n.fetch_add(1, std::memory_order_relaxed);
  if (i < kMax) g_ip[i] = (unsigned long)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
}
__attribute__((constructor)) static void start() {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_prof;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, nullptr);
  struct itimerval it = {{0, 1000}, {0, 1000}};
  setitimer(ITIMER_PROF, &it, nullptr);
}
__attribute__((destructor)) static void stop() {
  struct itimerval it = {{0, 0}, {0, 0}};
  setitimer(ITIMER_PROF, &it, nullptr);
  unsigned long lo = 0, hi = 0;
  FILE* m = fopen("/proc/self/maps", "r");
  char line[512];
  while (m && fgets(line, sizeof line, m)) {
    if (!strstr(line, "libp3host.so")) continue;
    unsigned long a, b;
    if (sscanf(line, "%lx-%lx", &a, &b) == 2) {
      if (!lo || a < lo) lo = a;
      if (b > hi) hi = b;
    }
  }
  if (m) fclose(m);
  int n = g_n.load();
  if (n > kMax) n = kMax;
  int inside = 0;
  for (int i = 0; i < n; ++i) inside += g_ip[i] >= lo && g_ip[i] < hi;
  if (!inside) return;   // a process that never ran the host (the launcher, helpers)
  const char* out = getenv("HOSTPROF_OUT");
  FILE* f = fopen(out ? out : "/tmp/hostprof.txt", "w");
  if (!f) return;
  for (int i = 0; i < n; ++i)
    if (g_ip[i] >= lo && g_ip[i] < hi) fprintf(f, "0x%lx\n", g_ip[i] - lo);
  fprintf(stderr, "hostprof: %d samples, %d in libp3host.so\n", n, inside);
  fclose(f);
}
