/* Entry points that exist ONLY in the laboratory build of the library (tools/lab/build_lab.sh -> tools/_build/libhgn_mp_lab.so,
 * selected with HGN_LIB=<that file>): measured-and-rejected kernel variants kept for comparison.  Not part of the C ABI of
 * include/hgn_mp.h; the shipped libhgn_mp.so does not export them. */
#pragma once
#include "hgn_mp.h"
#ifdef __cplusplus
extern "C" {
#endif
/* 1 if the arguments have the edge-block shape of the weight-stationary forward (csrc/ws_fwd.hip: one 128-wide ungathered source,
 * <= 2 gathered pre-projections, LayerNorm, packed weights, six products): the kernel hgn_mlp_fwd then takes for it. */
int hgn_mlp_fwd_ws_eligible(const hgn_mlp_fwd_t* args /*host*/);
/* Diagnostic (initial value: environment HGN_BIG_TILES set): 1 = forward launches of >= 98 304 rows run the split-bf16 MLP kernel as
 * 12-wave workgroups on 192-row tiles (one 96 KB weight stage per CU instead of three 48 KB ones).  Same results bit for bit, same speed. */
int hgn_set_big_tiles(int on);
int hgn_set_ws_fwd(int on);        /* process-wide switch of that kernel (initial value: environment HGN_WS_FWD set) */
#ifdef __cplusplus
}
#endif
