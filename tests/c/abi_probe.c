/* Plain-C consumer of the drop-in boundary: includes both public headers as C11, takes the address of one entry point per group and
 * calls the ones that need no device.  Built and run by tests/test_abi.py (gcc, no HIP headers): what a cgo / bindgen binding sees. */
#include <stdio.h>
#include <string.h>

#include "dfgpu.h"
#include "dfgpu_exec.h"

int main(void) {
  /* typed pointers: a signature drifting from the header no longer compiles */
  dfgpu_status (*p_probe)(dfgpu_ctx *, const dfgpu_join_table *, const dfgpu_array *const *, int32_t, const dfgpu_array *, dfgpu_array **, dfgpu_array **) = dfgpu_join_probe;
  dfgpu_status (*p_intern)(dfgpu_ctx *, dfgpu_groups *, const dfgpu_array *const *, int32_t, const dfgpu_array *, dfgpu_array **) = dfgpu_groups_intern;
  dfgpu_status (*p_update)(dfgpu_ctx *, dfgpu_acc *, const dfgpu_array *, const dfgpu_array *, const dfgpu_array *, int64_t) = dfgpu_acc_update_batch;
  dfgpu_status (*p_take)(dfgpu_ctx *, const dfgpu_array *, const dfgpu_array *, dfgpu_array **) = dfgpu_take;
  if (!p_probe || !p_intern || !p_update || !p_take) return 2;
  dfgpu_expr_node node = { DFGPU_NODE_COLUMN, 0, 0 };
  if (node.op != -1) return 3;
  /* no device in this process' environment is an error status, never a crash or a CPU fallback */
  dfgpu_ctx *ctx = NULL;
  dfgpu_status st = dfgpu_ctx_create(0, NULL, &ctx);
  if (st == DFGPU_OK && ctx) { printf("ctx ok\n"); dfgpu_ctx_destroy(ctx); }
  else printf("ctx status %d\n", (int)st);
  /* the run-time compiled kernel text compiles without a device */
  char log[4096]; log[0] = 0;
  st = dfgpu_jit_selftest("gfx950", log, (int64_t)sizeof log);
  printf("jit selftest %d\n", (int)st);
  return st == DFGPU_OK ? 0 : 4;
}
