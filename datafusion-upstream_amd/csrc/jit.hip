// jit.hip -- run-time compiled kernels (hiprtc) for row pipelines whose shape is only known when the plan runs: the fused
// "evaluate aggregate arguments + accumulate" pass of acc.hip.  The kernel TEXT is written by hand (a fixed loop around a generated
// straight-line expression body); compiling it for the query at hand keeps every partial sum in registers, which an interpreter over a
// per-lane register file cannot do.  One compile per distinct source text per process (tens of ms after the first).
#include <hip/hiprtc.h>

#include <map>
#include <mutex>

#include "dfgpu_internal.h"

namespace dfgpu {

namespace {
struct Compiled { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; };
std::mutex g_jit_mu;
std::map<std::string, Compiled> g_jit;      // key: device arch + '\n' + source
}  // namespace

void* jit_kernel(dfgpu_ctx* ctx, const std::string& source, const char* name) {
  hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
  std::string arch = prop.gcnArchName;
  std::string key = arch + "\n" + source;
  std::lock_guard<std::mutex> l(g_jit_mu);
  auto it = g_jit.find(key);
  if (it != g_jit.end()) return (void*)it->second.fn;
  hiprtcProgram prog;
  // A compiler that cannot be used (library missing its code-object manager, ...) is not an error of the query: the caller has
  // precompiled kernels for the same work, so the fused entry point answers NotImplemented -- once, loudly on stderr.
  if (hiprtcCreateProgram(&prog, source.c_str(), "dfgpu_fused.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { fprintf(stderr, "dfgpu: hiprtcCreateProgram failed; fused kernels disabled for this call\n"); fail(DFGPU_NOT_IMPLEMENTED, "hiprtcCreateProgram failed"); }
  std::string archopt = "--offload-arch=" + arch;
  const char* opts[] = {archopt.c_str(), "-O3", "-munsafe-fp-atomics"};
  hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls); std::string log(ls, '\0'); if (ls) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    fprintf(stderr, "dfgpu: hiprtc could not compile a fused kernel (%s); the node-by-node kernels run instead\n%.1500s\n", hiprtcGetErrorString(r), log.c_str());
    fail(DFGPU_NOT_IMPLEMENTED, "hiprtc: %s", hiprtcGetErrorString(r));
  }
  size_t cs = 0; hiprtcGetCodeSize(prog, &cs); std::vector<char> code(cs); hiprtcGetCode(prog, code.data()); hiprtcDestroyProgram(&prog);
  Compiled c;
  HIP_CHECK(hipModuleLoadData(&c.mod, code.data()));
  HIP_CHECK(hipModuleGetFunction(&c.fn, c.mod, name));
  g_jit.emplace(std::move(key), c);          // modules live as long as the process: plans re-run with the same text
  return (void*)c.fn;
}

// compile only (no device needed): the build check of __graft_entry__.build() runs the generator's output through the same compiler
bool jit_compile_only(const std::string& source, const char* arch, std::string* log) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, source.c_str(), "dfgpu_fused.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { *log = "hiprtcCreateProgram failed"; return false; }
  std::string archopt = std::string("--offload-arch=") + arch;
  const char* opts[] = {archopt.c_str(), "-O3", "-munsafe-fp-atomics"};
  hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls); log->assign(ls, '\0'); if (ls) hiprtcGetProgramLog(prog, &(*log)[0]);
  hiprtcDestroyProgram(&prog);
  return r == HIPRTC_SUCCESS;
}

}  // namespace dfgpu
