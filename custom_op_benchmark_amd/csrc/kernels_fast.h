// Fast fp32 kernels for gfx950 (wave64), by family:
//   kernels_base.h     row / stream loads, per-row slot-range loops, float-atomic row flushes
//   kernels_chunk.h    CHUNK drivers k_sddmm_f32 / k_spmm_f32 (any chunk layout, no plan)
//   kernels_strip.h    strip inner loops of the column-window drivers, staged id streams (IdStage)
//   kernels_wown.h     WINDOW-OWNER drivers k_sddmm_wown*_f32 / k_spmm_wown*_f32 (plan; XCDs own L2 windows)
//   kernels_nme.h      node_mul_edge streaming kernels
//   kernels_softmax.h  row-segment softmax / its backward
// (kernels_walk.h -- the WALK drivers -- includes this header; kernels_block.h, kernels_attn.h, kernels_generic.h
// are the block-dense MFMA, fused-attention and any-shape / fp64 families.)
#pragma once
#include "kernels_base.h"
#include "kernels_chunk.h"
#include "kernels_strip.h"
#include "kernels_wown.h"
#include "kernels_nme.h"
#include "kernels_softmax.h"
