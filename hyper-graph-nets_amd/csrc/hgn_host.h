// Host-side helpers shared by the translation units of libhgn_mp.so (error reporting, optional profiler).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hgn_mp.h"

namespace hgn {

int hgn_fail(int code, const char* msg);          // records msg (thread-local) and returns code
int hgn_check_launch(const char* what);
int matmul_products();                            // 6 (fp32-accurate split products) or 1 (single bf16 product): hgn_set_matmul_products
extern thread_local int g_prof_tag;           // hipGetLastError() -> HGN_OK / HGN_E_LAUNCH

// Records a HIP event pair around the launches issued in its scope when profiling is enabled.
struct ProfScope {
  int kid; hipStream_t stream; bool on; int slot;
  ProfScope(int kernel_id, double units, hipStream_t s);
  ~ProfScope();
};

}  // namespace hgn
