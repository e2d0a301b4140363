// Backward of the field query (nfi_field_query_bwd and the binned plane-gradient scatter) as its own translation unit.
//
// Built with -fno-slp-vectorize: with LLVM's SLP vectoriser on, the coordinate-gradient arithmetic of
// field_query_bwd_kernel is emitted as packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 with cross-half op_sel),
// and on MI355X the product fa * (dc3 - dc1) of the LAST plane then came out as zero for the wave's lanes 48..63 once
// in about 1e5 tiles, depending on timing (round 2; isolated with debug outputs of the partial terms: the inputs and the
// other half of the packed result were right; tools/determinism_probe.py counts the events).  Without packed fp32 in this kernel: 0 events in 3000 launches against 54 in 1500, and the
// kernel is not slower ("Determinism of the backward").
#include "nfi_host.hpp"

#include <algorithm>
#include <cstring>

using namespace nfi;

#include "nfi_backward_field.inc"
