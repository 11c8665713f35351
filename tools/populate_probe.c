// Does MADV_POPULATE_WRITE work on this box, and what does mapping 1 GiB cost (huge pages asked for)?
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(void) {
  const size_t n = 1ul << 30;
  for (int mode = 0; mode < 3; ++mode) {
    char *p = mmap(NULL, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    char *a = (char *)(((unsigned long)p + (2 << 20) - 1) & ~((2ul << 20) - 1));
    int r1 = madvise(a, n, MADV_HUGEPAGE), e1 = errno;
    double t0 = now();
    int r2 = 0, e2 = 0;
    if (mode == 0) { r2 = madvise(a, n, MADV_POPULATE_WRITE); e2 = errno; }
    else if (mode == 1) { for (size_t i = 0; i < n; i += 4096) a[i] = 1; }
    else { for (size_t i = 0; i < n; i += (2 << 20)) a[i] = 1; }
    double t1 = now();
    printf("mode %d (%s): hugepage rc %d (%s), populate rc %d (%s), %.1f ms for 1 GiB\n", mode,
           mode == 0 ? "MADV_POPULATE_WRITE" : mode == 1 ? "touch every 4 KiB" : "touch every 2 MiB", r1, r1 ? strerror(e1) : "ok", r2,
           r2 ? strerror(e2) : "ok", (t1 - t0) * 1e3);
    munmap(p, n + (2 << 20));
  }
  FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char buf[128] = {0};
  if (f) { fgets(buf, sizeof buf, f); fclose(f); }
  printf("THP: %s", buf);
  return 0;
}
