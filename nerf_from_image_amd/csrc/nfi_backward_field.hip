// Backward of the field query (nfi_field_query_bwd and the binned plane-gradient scatter) as its own translation unit.
//
// Built with -fno-slp-vectorize (__graft_entry__.UNITS): measured faster for this unit (2.66 vs 2.79 ms per 4.2 M points).
// Round 2 had switched the vectoriser off because the packed-fp32 code it emits for the coordinate gradients came out wrong
// once in ~1e5 tiles; round 3 found the cause - a gfx950 fault of one VOP3P operand form next to a K = 32 16-bit MFMA
// (tools/probes/pk_hazard.hip) - and removes that form from BOTH units' assembly at build time
// (tools/gfx950_pk_legalize.py, HISTORY.md "Determinism: root cause"), so the flag is a speed choice now, not a fix.
#include "nfi_host.hpp"

#include <algorithm>
#include <cstring>

using namespace nfi;

#include "nfi_backward_field.inc"
