// Probe: does hiprtc compile a gfx950 kernel in this image, and how long does it take?
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>
int main() {
  const char* src = R"(
extern "C" __global__ void k(const double* a, const double* b, double* o, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) o[i] = a[i] * (1.0 - b[i]);
})";
  auto t0 = std::chrono::steady_clock::now();
  hiprtcProgram p; if (hiprtcCreateProgram(&p, src, "k.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { puts("create failed"); return 1; }
  const char* opts[] = {"--offload-arch=gfx950", "-O3"};
  hiprtcResult r = hiprtcCompileProgram(p, 2, opts);
  size_t ls = 0; hiprtcGetProgramLogSize(p, &ls); std::string log(ls, 0); if (ls) hiprtcGetProgramLog(p, &log[0]);
  if (r != HIPRTC_SUCCESS) { printf("compile failed: %s\n%s\n", hiprtcGetErrorString(r), log.c_str()); return 1; }
  size_t cs = 0; hiprtcGetCodeSize(p, &cs);
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  printf("ok code=%zu bytes in %.0f ms\n", cs, ms);
  return 0;
}
