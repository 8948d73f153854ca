// Library-wide tuning / test knobs: one table of plain ints the launch paths read; changed at run time through smx_set_tuning (an explicit call:
// tests, tools).  The SHIPPED library never reads the environment -- a product run's kernel selection cannot be changed by a stray variable;
// a -DSMX_TOOLS build (tools/*.py, tools/*.sh: A/B timing, ablations) also initialises the table ONCE from SMX_* variables.
// -1 = "auto" (the launcher's own device-tuned choice).
#include <string.h>
#include <stdlib.h>
#include "smx.h"
#include "smx_common.h"

namespace {
struct Knob { const char* name; const char* env; int value; };
Knob g_knobs[SMX_TUNE_COUNT] = {
  {"wino_nw", "SMX_WINO_NW", -1},               // Winograd: 32-wide N tiles per block (1 | 2)
  {"wino_ablate", "SMX_WINO_ABLATE", 0},        // Winograd: timing-only ablation mask (tools)
  {"gemm_variant", "SMX_GEMM_VARIANT", 3},      // implicit GEMM: bit0 store-early pipeline, bit1 s_setprio
  {"gemm_xcd_swizzle", "SMX_GEMM_XCD_SWIZZLE", 1},
  {"warp_rows", "SMX_WARP_ROWS", 1},            // warp: row-chunk kernel when eligible (0 = per-lane kernel)
  {"warp_reorder", "SMX_WARP_REORDER", 1},      // warp: (XCD, chunk, frame) block order
  {"attn16", "SMX_ATTN16", 1},                  // bf16 storage, d_head 32: bf16 MFMA kernel (0 = fp32 MFMA kernel on bf16 storage)
  {"wino_wide", "SMX_WINO_WIDE", 1},            // Winograd big launches: 1 = 4-wave "wide" blocks, 64 n per wave (default), 0 = 8-wave blocks, 5 = wide at one block per CU (tools)
  {"wino_nt", "SMX_WINO_NT", 0},                // wide Winograd epilogue: non-temporal residual loads / output stores
  {"attn4_mfma", "SMX_ATTN4_MFMA", 1},          // d_head 4 attention: 1 = MFMA kernels (fp32 storage: 4x4x1 f32; bf16 storage: 4x4x4 bf16), 2 = the 4x4x1 f32 kernel on either storage, 0 = VALU kernel
  {"conv16_slab", "SMX_CONV16_SLAB", 1},        // bf16 region conv, 16x16 tiles: 1 = 32-channel slices with all nine taps' weights in LDS, 0 = one weight tile per tap
  {"attn_bwd_mfma", "SMX_ATTN_BWD_MFMA", 1},    // training, d_head 32 attention backward: 1 = fp32 MFMA kernels, 0 = the per-thread VALU kernels
  {"vq_split", "SMX_VQ_SPLIT", 1},              // VQ, few tokens (< half a chip of 128-token blocks): codebook sweep split over blockIdx.y + a combine kernel
  {"warp_nt", "SMX_WARP_NT", 1},                // warp row-chunk kernels: non-temporal output stores (write-once stream: 6.5 -> 8.3 TB/s algorithmic at B = 300; 0 = plain stores)
  {"wgrad_region", "SMX_WGRAD_REGION", 1},      // training: 3x3 / s1 / p1 weight gradients with 64-multiple channels on the region kernel (0 = the generic TN GEMM)
  {"wgrad_slots", "SMX_WGRAD_SLOTS", 256},      // region weight gradient: blocks the pixel split aims at (device sweep profiles/r04_wgrad_region.txt: 256 = one block per CU beats 512 -- half the partials to write and reduce, twice the rows per run -- and 128)
  {"wino_stagger", "SMX_WINO_STAGGER", 0},      // wide Winograd: shader cycles the second resident round of a launch's first blocks waits (phase stagger of the two blocks of a CU); 0 = off
  {"wino_ws", "SMX_WINO_WS", 0},                // TOOLS BUILD ONLY: wide Winograd launches on the producer / consumer-split kernel (winograd_ws_kernel: persistent 8-wave blocks, MFMA waves + helper waves); ignored by the shipped library
  {"gemm_loader", "SMX_GEMM_LOADER", 1},        // gemm_conv: 1 = operand quads by buffer loads with SGPR bases / offsets where the layout allows (MODE 2), 0 = the float4 gather
  {"wino_xcd", "SMX_WINO_XCD", 1},              // wide Winograd: the output blocks of a spatial tile on ONE XCD at the same time (block ids regrouped 8 apart): its input region comes from HBM once
  {"wino_bf3_shape", "SMX_WINO_BF3_SHAPE", -1}, // split-bf16 Winograd block shape: -1 auto (8x16 pixels x 128 channels when C_out % 128 == 0), 2 = always 16x16 pixels x 64 channels
  {"attn_bf3", "SMX_ATTN_BF3", 4},                // fp32 storage, d_head 32, >= 512 blocks: split attention (4 = f16x3: two half levels, three products; 3 = bf16 six products everywhere, 2 = bf16 with P on two levels, + 16 = at any launch size, + 32 = 128-query blocks instead of 256, 0 = the fp32-MFMA kernel)
};
bool g_init = false;
void init_once() {
  if (g_init) return;
  g_init = true;
#ifdef SMX_TOOLS
  for (int i = 0; i < SMX_TUNE_COUNT; ++i) {
    const char* e = getenv(g_knobs[i].env);
    if (e && *e) g_knobs[i].value = atoi(e);
  }
#endif
}
}  // namespace

int smx_tune(int key) {
  init_once();
  return (key >= 0 && key < SMX_TUNE_COUNT) ? g_knobs[key].value : -1;
}

extern "C" int smx_set_tuning(const char* name, int value) {
  init_once();
  if (!name) return SMX_EINVAL;
  for (int i = 0; i < SMX_TUNE_COUNT; ++i)
    if (!strcmp(name, g_knobs[i].name)) { g_knobs[i].value = value; return SMX_OK; }
  return SMX_EINVAL;
}

extern "C" int smx_get_tuning(const char* name, int* value) {
  init_once();
  if (!name || !value) return SMX_EINVAL;
  for (int i = 0; i < SMX_TUNE_COUNT; ++i)
    if (!strcmp(name, g_knobs[i].name)) { *value = g_knobs[i].value; return SMX_OK; }
  return SMX_EINVAL;
}
