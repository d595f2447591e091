// How long do hipMalloc / hipFree of tens of GB take on this stack?  (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  (void)hipFree(0);
  for (int rep = 0; rep < 3; rep++)
    for (size_t gb : {1, 8, 60, 120}) {
      void* p = nullptr;
      double t0 = now();
      hipError_t e = hipMalloc(&p, gb << 30);
      double t1 = now();
      if (e != hipSuccess) { printf("%zu GB: %s\n", gb, hipGetErrorString(e)); continue; }
      (void)hipMemset(p, 0, gb << 30);
      (void)hipDeviceSynchronize();
      double t2 = now();
      (void)hipFree(p);
      double t3 = now();
      printf("rep %d: %3zu GB  hipMalloc %.3f s  memset %.3f s  hipFree %.3f s\n", rep, gb, t1 - t0, t2 - t1, t3 - t2);
    }
  return 0;
}
