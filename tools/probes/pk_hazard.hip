// probe: timing-dependent wrong results around packed fp32 (v_pk_*_f32), DPP and cross-half op_sel on gfx950.
//
// Background (HISTORY.md "Determinism"): field_query_bwd_kernel, when its coordinate-gradient arithmetic is compiled by
// the SLP vectoriser to v_pk_mul_f32 / v_pk_add_f32 with op_sel, produced a wrong g_points component in lanes 48..63 once
// in ~1e5 tiles.  This probe runs instruction sequences TWICE per iteration from the same inputs - once exactly as the
// compiler scheduled them ("tight") and once with s_nop 7 between all instructions ("padded") - and compares the
// results bit for bit, while the same wave keeps global loads, LDS reads and MFMAs in flight and a second wave shares
// the SIMD (2 waves / SIMD as in the kernel).  Inline asm is opaque to the compiler's hazard recogniser, so the wait
// states are exactly the ones written here.
//
//   test 0  literal replica of the compiled tail of the coordinate-gradient block (register numbers as in the ISA of
//           field_query_bwd_kernel<true,true,false,0> built WITHOUT -fno-slp-vectorize), variants:
//             0 tight   1 nop after producers of DPP sources   2 nop after the DPP moves   3 nop between the
//             differences and the op_sel products   4 nop after the products   5 / 6 only s_nop 0 / s_nop 1 there
//             7 s_nop 0 between all packed instructions
//   test 1  v_pk_add_f32 -> N wait states -> v_mov_b32_dpp of the HIGH and LOW result halves (N = variant, 0..4;
//           LLVM inserts 2 = the documented VALU-write -> DPP-read requirement)
//   test 2  same with a plain v_add_f32 producer (control)
//   test 3  v_mov_b32_dpp -> v_pk_add_f32 consumer back to back
//   test 4  v_pk_add_f32 (neg) -> v_pk_mul_f32 with cross-half op_sel in place -> v_pk_add_f32 consumer back to back
//
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_hazard.hip -o tools/probes/pk_hazard ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kIn = 28, kOut = 8;
constexpr int kStride = 1024;  // bytes between consecutive per-thread slots (256 threads x 4 B)

// ---- LDS slot helpers inside the asm: input k at offset k*1024, outputs behind the inputs ----
#define LDI(reg, k) "ds_read_b32 " reg ", %[a] offset:" #k "*1024\n"
#define STO(reg, k) "ds_write_b32 %[a], " reg " offset:(28+" #k ")*1024\n"
#define STP(reg, k) "ds_write_b32 %[a], " reg " offset:(36+" #k ")*1024\n"
#define WAITL "s_waitcnt lgkmcnt(0)\ns_nop 7\n"
#define N7 "s_nop 7\n"

// ---- test 0: the literal block; S_P: after plain VALU producers of DPP sources, S_D: after DPP moves, S_M: between the
// differences and the op_sel products, S_A: after the products, S_X: everywhere else ----
#define T0_LOAD \
  LDI("v6", 0) LDI("v7", 1) LDI("v8", 2) LDI("v10", 3) LDI("v11", 4) LDI("v68", 5) LDI("v69", 6) LDI("v198", 7)        \
  LDI("v199", 8) LDI("v200", 9) LDI("v201", 10) LDI("v204", 11) LDI("v205", 12) LDI("v206", 13) LDI("v207", 14)        \
  LDI("v210", 15) LDI("v211", 16) LDI("v78", 17) LDI("v79", 18) LDI("v126", 19) LDI("v127", 20) LDI("v128", 21)        \
  LDI("v129", 22) LDI("v130", 23) LDI("v131", 24) LDI("v132", 25) LDI("v133", 26) "v_mov_b32 v9, 0\n" WAITL
#define T0_SEQ(S_P, S_D, S_M, S_A, S_X)                                                                            \
  "v_pk_add_f32 v[54:55], v[198:199], v[200:201]\n" S_X                                                           \
  "v_mov_b32_e32 v208, v9\n" "v_mov_b32_e32 v209, v9\n" "v_mov_b32_e32 v212, v9\n" "v_mov_b32_e32 v213, v9\n"      \
  "v_mov_b32_e32 v80, v9\n" "v_mov_b32_e32 v81, v9\n" S_X                                                         \
  "v_pk_fma_f32 v[10:11], v[8:9], v[68:69], v[10:11] op_sel_hi:[0,1,1]\n" S_X                                     \
  "v_mov_b32_e32 v12, v9\n" "v_mov_b32_e32 v13, v9\n" S_X                                                         \
  "v_pk_add_f32 v[6:7], v[54:55], v[6:7] op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]\n" S_X            \
  "v_pk_add_f32 v[54:55], v[204:205], v[54:55] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n" S_P                   \
  "v_mov_b32_dpp v208, v206 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                     \
  "v_mov_b32_dpp v209, v207 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                     \
  "v_mov_b32_dpp v212, v210 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                     \
  "v_mov_b32_dpp v213, v211 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                     \
  "v_mov_b32_dpp v80, v78 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v81, v79 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v12, v10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v13, v11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" S_D                                   \
  "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n" S_X                                                                 \
  "v_pk_mul_f32 v[54:55], v[54:55], v[126:127]\n" S_X                                                             \
  "v_mov_b32_e32 v56, v9\n" S_X                                                                                   \
  "v_pk_add_f32 v[6:7], v[54:55], v[6:7]\n" S_X                                                                   \
  "v_pk_add_f32 v[54:55], v[206:207], v[208:209]\n" S_X                                                           \
  "v_pk_add_f32 v[58:59], v[210:211], v[212:213]\n" S_X                                                           \
  "v_mov_b32_e32 v60, v9\n" S_X                                                                                   \
  "v_pk_add_f32 v[62:63], v[78:79], v[80:81]\n" S_X                                                               \
  "v_mov_b32_e32 v64, v9\n" S_X                                                                                   \
  "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n" S_X                                                               \
  "v_mov_b32_e32 v12, v9\n" "v_mov_b32_e32 v57, v9\n" "v_mov_b32_e32 v65, v9\n" "v_mov_b32_e32 v61, v9\n"          \
  "v_mov_b32_e32 v13, v9\n" S_P                                                                                   \
  "v_mov_b32_dpp v56, v54 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v60, v58 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v64, v62 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v12, v10 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v57, v55 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v65, v63 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v61, v59 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                                       \
  "v_mov_b32_dpp v13, v11 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" S_D                                   \
  "v_pk_add_f32 v[54:55], v[54:55], v[56:57]\n" S_X                                                               \
  "v_pk_add_f32 v[56:57], v[62:63], v[64:65]\n" S_X                                                               \
  "v_pk_add_f32 v[58:59], v[58:59], v[60:61]\n" S_X                                                               \
  "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n" S_X                                                               \
  "v_pk_add_f32 v[12:13], v[58:59], v[54:55] neg_lo:[0,1] neg_hi:[0,1]\n" S_X                                     \
  "v_pk_add_f32 v[60:61], v[10:11], v[56:57] neg_lo:[0,1] neg_hi:[0,1]\n" S_X                                     \
  "v_pk_add_f32 v[54:55], v[56:57], v[54:55] neg_lo:[0,1] neg_hi:[0,1]\n" S_X                                     \
  "v_pk_add_f32 v[10:11], v[10:11], v[58:59] neg_lo:[0,1] neg_hi:[0,1]\n" S_M                                     \
  "v_pk_mul_f32 v[54:55], v[128:129], v[54:55] op_sel:[1,0] op_sel_hi:[0,1]\n" S_X                                \
  "v_pk_mul_f32 v[10:11], v[10:11], v[126:127] op_sel:[0,1] op_sel_hi:[1,0]\n" S_X                                \
  "v_pk_mul_f32 v[12:13], v[132:133], v[12:13] op_sel_hi:[0,1]\n" S_X                                             \
  "v_pk_mul_f32 v[60:61], v[60:61], v[130:131] op_sel_hi:[1,0]\n" S_A                                             \
  "v_pk_add_f32 v[10:11], v[10:11], v[54:55]\n" S_X                                                               \
  "v_pk_add_f32 v[6:7], v[6:7], 0 op_sel_hi:[1,0]\n" S_X                                                          \
  "v_pk_add_f32 v[12:13], v[12:13], v[60:61]\n" S_X                                                               \
  "v_add_f32_e32 v8, 0, v10\n" S_X                                                                                \
  "v_pk_add_f32 v[6:7], v[6:7], v[12:13]\n" S_X                                                                   \
  "v_add_f32_e32 v8, v8, v11\n" N7
#define T0_STORE(ST) ST("v6", 0) ST("v7", 1) ST("v8", 2) ST("v10", 3) ST("v11", 4) ST("v12", 5) ST("v13", 6) ST("v54", 7) "s_waitcnt lgkmcnt(0)\n"
#define T0_CLOB "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62",  \
                "v63", "v64", "v65", "v68", "v69", "v78", "v79", "v80", "v81", "v126", "v127", "v128", "v129", "v130", "v131",     \
                "v132", "v133", "v198", "v199", "v200", "v201", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211",  \
                "v212", "v213", "memory"

// ---- tests 1-4: small sequences on v[200..215] ----
#define TS_LOAD LDI("v200", 0) LDI("v201", 1) LDI("v202", 2) LDI("v203", 3) LDI("v208", 4) LDI("v209", 5) LDI("v212", 6) LDI("v213", 7) \
  "v_mov_b32 v206, 0\nv_mov_b32 v207, 0\nv_mov_b32 v204, 0\nv_mov_b32 v205, 0\nv_mov_b32 v210, 0\nv_mov_b32 v211, 0\n" WAITL
#define TS_STORE(ST) ST("v204", 0) ST("v205", 1) ST("v206", 2) ST("v207", 3) ST("v210", 4) ST("v211", 5) ST("v200", 6) ST("v201", 7) "s_waitcnt lgkmcnt(0)\n"
#define TS_CLOB "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "memory"
// 1: packed producer -> W -> DPP reads of both halves
#define T1_SEQ(W) "v_pk_add_f32 v[204:205], v[200:201], v[202:203]\n" W                                           \
  "v_mov_b32_dpp v207, v205 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                     \
  "v_mov_b32_dpp v206, v204 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" N7
// 2: scalar producer -> W -> DPP read
#define T2_SEQ(W) "v_add_f32_e32 v205, v201, v203\n" "v_add_f32_e32 v204, v200, v202\n" W                           \
  "v_mov_b32_dpp v206, v204 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                                     \
  "v_mov_b32_dpp v207, v205 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" N7
// 3: DPP moves -> W -> packed consumer
#define T3_SEQ(W) "v_mov_b32_dpp v206, v200 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                      \
  "v_mov_b32_dpp v207, v201 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" W                                   \
  "v_pk_add_f32 v[204:205], v[200:201], v[206:207]\n" W                                                           \
  "v_pk_add_f32 v[210:211], v[204:205], v[202:203] neg_lo:[0,1] neg_hi:[0,1]\n" N7
// 4: packed difference -> cross-half op_sel product in place -> packed consumer
#define T4_SEQ(W) "v_pk_add_f32 v[204:205], v[200:201], v[202:203] neg_lo:[0,1] neg_hi:[0,1]\n" W                   \
  "v_pk_mul_f32 v[204:205], v[204:205], v[208:209] op_sel:[0,1] op_sel_hi:[1,0]\n" W                              \
  "v_pk_mul_f32 v[206:207], v[212:213], v[200:201] op_sel:[1,0] op_sel_hi:[0,1]\n" W                              \
  "v_pk_add_f32 v[210:211], v[204:205], v[206:207]\n" N7

// 5: ONE instruction between s_nop 7 pads, result d = v[204:205] (form 5: in place, v[200:201]); a = v[200:201], b = v[202:203], c = v[208:209]
#define T5_SEQ(INSTR) N7 INSTR N7
#define F0 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203]\n"
#define F1 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel:[0,1] op_sel_hi:[1,0]\n"
#define F2 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel:[1,0] op_sel_hi:[0,1]\n"
#define F3 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel_hi:[0,1]\n"
#define F4 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel_hi:[1,0]\n"
#define F5 "v_pk_mul_f32 v[200:201], v[200:201], v[202:203] op_sel:[0,1] op_sel_hi:[1,0]\n"
#define F6 "v_pk_add_f32 v[204:205], v[200:201], v[202:203]\n"
#define F7 "v_pk_add_f32 v[204:205], v[200:201], v[202:203] op_sel:[0,1] op_sel_hi:[1,0]\n"
#define F8 "v_pk_fma_f32 v[204:205], v[200:201], v[202:203], v[208:209]\n"
#define F9 "v_pk_fma_f32 v[204:205], v[200:201], v[202:203], v[208:209] op_sel:[1,0,0]\n"
#define F10 "v_pk_fma_f32 v[204:205], v[200:201], v[202:203], v[208:209] op_sel_hi:[0,1,1]\n"
#define F11 "v_mul_f32_e32 v204, v200, v202\nv_mul_f32_e32 v205, v201, v203\n"
#define F12 "v_fma_f32 v204, v200, v202, v208\nv_fma_f32 v205, v201, v203, v209\n"
#define F13 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel:[1,1] op_sel_hi:[0,0]\n"
#define F14 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel:[0,1]\n"
#define F15 "v_pk_fma_f32 v[204:205], v[200:201], v[202:203], v[208:209] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
#define F16 "v_pk_fma_f32 v[204:205], v[200:201], v[202:203], v[208:209] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n"
#define F17 "v_pk_fma_f32 v[204:205], v[200:201], v[202:203], v[208:209] op_sel:[1,0,0] op_sel_hi:[0,1,1]\n"
#define F18 "v_pk_mov_b32 v[204:205], v[200:201], v[202:203] op_sel:[0,1]\n"
#define F19 "v_pk_mov_b32 v[204:205], v[200:201], v[202:203] op_sel:[1,0]\n"
#define F20 "v_pk_add_f32 v[204:205], v[200:201], v[202:203] op_sel:[0,1] op_sel_hi:[0,0]\n"
#define F21 "v_pk_add_f32 v[204:205], v[200:201], v[202:203] op_sel:[1,1] op_sel_hi:[1,0]\n"
#define F22 "v_pk_add_f32 v[204:205], v[200:201], v[200:201] op_sel:[0,1] op_sel_hi:[1,0]\n"
#define F23 "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"

struct Params {
  unsigned* err;          // [tests][variants][5]: mismatches in lane quarters 0..3, total iterations with a mismatch
  float* examples;        // first mismatches: [64][4] = {test*16+variant, lane, tight, padded}
  unsigned* n_examples;
  const f32x4* traffic; unsigned traffic_len;
  float* sink;
  int iters;
  int slot;               // which err[] record this launch fills
};

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int TEST, int VAR>
__device__ __forceinline__ void run_pair(unsigned a) {
  // tight (variant VAR) -> outputs at slots 28.., padded -> outputs at slots 36..
  if constexpr (TEST == 0) {
    if constexpr (VAR == 0) asm volatile(T0_LOAD T0_SEQ("", "", "", "", "") T0_STORE(STO) :: [a] "v"(a) : T0_CLOB);
    if constexpr (VAR == 1) asm volatile(T0_LOAD T0_SEQ(N7, "", "", "", "") T0_STORE(STO) :: [a] "v"(a) : T0_CLOB);
    if constexpr (VAR == 2) asm volatile(T0_LOAD T0_SEQ("", N7, "", "", "") T0_STORE(STO) :: [a] "v"(a) : T0_CLOB);
    if constexpr (VAR == 3) asm volatile(T0_LOAD T0_SEQ("", "", N7, "", "") T0_STORE(STO) :: [a] "v"(a) : T0_CLOB);
    if constexpr (VAR == 4) asm volatile(T0_LOAD T0_SEQ("", "", "", N7, "") T0_STORE(STO) :: [a] "v"(a) : T0_CLOB);
    if constexpr (VAR == 5) asm volatile(T0_LOAD T0_SEQ("", "", "s_nop 0\n", "", "") T0_STORE(STO) :: [a] "v"(a) : T0_CLOB);
    if constexpr (VAR == 6) asm volatile(T0_LOAD T0_SEQ("", "", "s_nop 1\n", "", "") T0_STORE(STO) :: [a] "v"(a) : T0_CLOB);
    if constexpr (VAR == 7) asm volatile(T0_LOAD T0_SEQ("", "", "", "", "s_nop 0\n") T0_STORE(STO) :: [a] "v"(a) : T0_CLOB);
    asm volatile(T0_LOAD T0_SEQ(N7, N7, N7, N7, N7) T0_STORE(STP) :: [a] "v"(a) : T0_CLOB);
  } else {
#define NFI_W(SEQ)                                                                                                 \
    if constexpr (VAR == 0) asm volatile(TS_LOAD SEQ("") TS_STORE(STO) :: [a] "v"(a) : TS_CLOB);                    \
    if constexpr (VAR == 1) asm volatile(TS_LOAD SEQ("s_nop 0\n") TS_STORE(STO) :: [a] "v"(a) : TS_CLOB);           \
    if constexpr (VAR == 2) asm volatile(TS_LOAD SEQ("s_nop 1\n") TS_STORE(STO) :: [a] "v"(a) : TS_CLOB);           \
    if constexpr (VAR == 3) asm volatile(TS_LOAD SEQ("s_nop 2\n") TS_STORE(STO) :: [a] "v"(a) : TS_CLOB);           \
    if constexpr (VAR == 4) asm volatile(TS_LOAD SEQ("s_nop 3\n") TS_STORE(STO) :: [a] "v"(a) : TS_CLOB);           \
    asm volatile(TS_LOAD SEQ(N7) TS_STORE(STP) :: [a] "v"(a) : TS_CLOB);
    if constexpr (TEST == 1) { NFI_W(T1_SEQ) }
    if constexpr (TEST == 2) { NFI_W(T2_SEQ) }
    if constexpr (TEST == 3) { NFI_W(T3_SEQ) }
    if constexpr (TEST == 4) { NFI_W(T4_SEQ) }
#undef NFI_W
#define NFI_F(K, F) if constexpr (VAR == K) asm volatile(TS_LOAD T5_SEQ(F) TS_STORE(STO) :: [a] "v"(a) : TS_CLOB);
    if constexpr (TEST == 5) {
      NFI_F(0, F0) NFI_F(1, F1) NFI_F(2, F2) NFI_F(3, F3) NFI_F(4, F4) NFI_F(5, F5) NFI_F(6, F6) NFI_F(7, F7) NFI_F(8, F8) NFI_F(9, F9)
      NFI_F(10, F10) NFI_F(11, F11) NFI_F(12, F12) NFI_F(13, F13) NFI_F(14, F14) NFI_F(15, F15) NFI_F(16, F16) NFI_F(17, F17)
      NFI_F(18, F18) NFI_F(19, F19) NFI_F(20, F20) NFI_F(21, F21) NFI_F(22, F22) NFI_F(23, F23)
    }
#undef NFI_F
  }
}

// PARTNER > 0: the block has 8 waves; waves 4..7 (the SIMD partners of the test waves 0..3) run a stream of ONE
// instruction class until the test waves are done: 1 v_mfma_f32_16x16x32_f16, 2 v_mfma_f32_16x16x4_f32,
// 3 v_permlane32_swap / v_permlane16_swap, 4 v_exp_f32 / v_log_f32, 5 DPP row operations, 6 ds_read_b128 / ds_write_b128,
// 7 buffer loads (global), 8 v_pk_fma_f32
template <int PARTNER>
__device__ __forceinline__ void partner_stream(const Params& p, volatile int* done, float* lds, unsigned gid) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (float)(gid & 255) + (float)i); hb[i] = (_Float16)0.5f; }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (float)(gid & 1023) + (float)i;
  unsigned it = 0;
  while (!*done) {
    ++it;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (PARTNER == 1) {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[a], 0, 0, 0);
      } else if constexpr (PARTNER == 9) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 a4 = {ha[0], ha[1], ha[2], ha[3]}, b4 = {hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[a], 0, 0, 0);
      } else if constexpr (PARTNER == 10) {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        const bf16x8 a8 = __builtin_bit_cast(bf16x8, ha), b8 = __builtin_bit_cast(bf16x8, hb);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[a], 0, 0, 0);
      } else if constexpr (PARTNER == 2) {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[a], v[4 + a], acc[a], 0, 0, 0);
      } else if constexpr (PARTNER == 3) {
#pragma unroll
        for (int a = 0; a < 8; a += 2) {
          auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[a]), __builtin_bit_cast(unsigned, v[a + 1]), false, false);
          auto r = __builtin_amdgcn_permlane16_swap(q[0], q[1], false, false);
          v[a] = __builtin_bit_cast(float, r[0]); v[a + 1] = __builtin_bit_cast(float, r[1]);
        }
      } else if constexpr (PARTNER == 4) {
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(v[a]));
      } else if constexpr (PARTNER == 5) {
#pragma unroll
        for (int a = 0; a < 8; ++a)
          v[a] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[a]), 0x111, 0xf, 0xf, false));
      } else if constexpr (PARTNER == 6) {
        f32x4* q = reinterpret_cast<f32x4*>(lds) + (threadIdx.x & 255);
        f32x4 x = q[0];
        x[0] += v[u & 7];
        q[256] = x;
        v[u & 7] = q[256][1] + x[2];
      } else if constexpr (PARTNER == 7) {
        const f32x4 t = p.traffic[(hash32(gid + it * 8 + u) >> 4) % p.traffic_len];
        v[u & 7] += t.x + t.w;
      } else if constexpr (PARTNER == 8) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int a = 0; a < 8; a += 2) {
          f32x2 x = {v[a], v[a + 1]}, y = {v[(a + 2) & 7], v[(a + 3) & 7]};
          x = __builtin_elementwise_fma(x, y, x);
          v[a] = x[0]; v[a + 1] = x[1];
        }
      }
    }
  }
  float sum = 0;
  for (int i = 0; i < 8; ++i) sum += v[i];
  for (int a = 0; a < 4; ++a) sum += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  p.sink[gid] = sum;
}

template <int TEST, int VAR, int PARTNER = 0>
__global__ __launch_bounds__(PARTNER ? 512 : 256) void probe(Params p) {
  __shared__ float slots[(kIn + 2 * kOut) * 256];
  __shared__ float noise[1024];
  __shared__ __attribute__((aligned(16))) float partner_lds[PARTNER == 6 ? 2 * 256 * 4 : 4];
  __shared__ int done_flag, done_count;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned gid = blockIdx.x * (PARTNER ? 512 : 256) + tid;
  for (int i = tid; i < 1024; i += 256) noise[i] = (float)i * 0.001f;
  if (tid == 0) { done_flag = 0; done_count = 0; }
  if (PARTNER == 6) for (int i = tid; i < 2 * 256 * 4; i += 512) partner_lds[i] = 0.5f;
  __syncthreads();
  if (PARTNER && wave >= 4) {
    partner_stream<PARTNER>(p, &done_flag, partner_lds, gid);
    return;
  }
  const unsigned a = (unsigned)(reinterpret_cast<uintptr_t>(slots) & 0xffffffffu) + tid * 4;   // LDS byte address of slot 0
  f32x4 acc = {0, 0, 0, 0};
  float sink = 0.0f;
  unsigned bad[4] = {0, 0, 0, 0}, bad_it = 0;
  for (int it = 0; it < p.iters; ++it) {
    const unsigned h = hash32(gid * 0x9e3779b9u + (unsigned)it);
#pragma unroll
    for (int k = 0; k < kIn; ++k) {
      const unsigned hk = hash32(h + 0x51ed27u * (k + 1));
      slots[k * 256 + tid] = __builtin_bit_cast(float, 0x3f000000u | (hk >> 9)) * ((hk & 1) ? 1.0f : -1.0f);   // +-[0.5, 1)
    }
    // interference kept in flight across the sequences: 4 global loads, 2 LDS reads, MFMAs on alternating turns
    const f32x4 t0 = p.traffic[(h >> 3) % p.traffic_len], t1 = p.traffic[(h >> 5) % p.traffic_len];
    const f32x4 t2 = p.traffic[(h >> 7) % p.traffic_len], t3 = p.traffic[(h >> 9) % p.traffic_len];
    const float l0 = noise[(h >> 4) & 1023], l1 = noise[(h >> 14) & 1023];
    if ((it + wave) & 1) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(l0, 1.0f, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(l1, 1.0f, acc, 0, 0, 0);
    }
    run_pair<TEST, VAR>(a);
    if constexpr (TEST == 5) {
      // expected values of the form, computed by the compiler's own scalar code from the same inputs
      const float a0 = slots[0 * 256 + tid], a1 = slots[1 * 256 + tid], b0 = slots[2 * 256 + tid], b1 = slots[3 * 256 + tid];
      const float c0 = slots[4 * 256 + tid], c1 = slots[5 * 256 + tid];
      float e0 = 0, e1 = 0;
      if (VAR == 0 || VAR == 11) { e0 = a0 * b0; e1 = a1 * b1; }
      if (VAR == 1 || VAR == 5) { e0 = a0 * b1; e1 = a1 * b0; }
      if (VAR == 2) { e0 = a1 * b0; e1 = a0 * b1; }
      if (VAR == 3) { e0 = a0 * b0; e1 = a0 * b1; }
      if (VAR == 4) { e0 = a0 * b0; e1 = a1 * b0; }
      if (VAR == 6) { e0 = a0 + b0; e1 = a1 + b1; }
      if (VAR == 7) { e0 = a0 + b1; e1 = a1 + b0; }
      if (VAR == 8 || VAR == 12) { e0 = __builtin_fmaf(a0, b0, c0); e1 = __builtin_fmaf(a1, b1, c1); }
      if (VAR == 9) { e0 = __builtin_fmaf(a1, b0, c0); e1 = __builtin_fmaf(a1, b1, c1); }
      if (VAR == 10) { e0 = __builtin_fmaf(a0, b0, c0); e1 = __builtin_fmaf(a0, b1, c1); }
      if (VAR == 13) { e0 = a1 * b1; e1 = a0 * b0; }
      if (VAR == 14) { e0 = a0 * b1; e1 = a1 * b1; }
      if (VAR == 15) { e0 = __builtin_fmaf(a0, b1, c0); e1 = __builtin_fmaf(a1, b0, c1); }
      if (VAR == 16) { e0 = __builtin_fmaf(a0, b0, c1); e1 = __builtin_fmaf(a1, b1, c0); }
      if (VAR == 17) { e0 = __builtin_fmaf(a1, b0, c0); e1 = __builtin_fmaf(a0, b1, c1); }
      if (VAR == 18) { e0 = a0; e1 = b1; }
      if (VAR == 19) { e0 = a1; e1 = b0; }
      if (VAR == 20) { e0 = a0 + b1; e1 = a0 + b0; }
      if (VAR == 21) { e0 = a1 + b1; e1 = a1 + b0; }
      if (VAR == 22) { e0 = a0 + a1; e1 = a1 + a0; }
      if (VAR == 23) { e0 = a0 * -b1; e1 = a1 * -b0; }
      const int o = VAR == 5 ? 6 : 0;            // in-place form: the result sits in v[200:201] = outputs 6, 7
#pragma unroll
      for (int k = 0; k < kOut; ++k) slots[(kIn + kOut + k) * 256 + tid] = slots[(kIn + k) * 256 + tid];
      slots[(kIn + kOut + o) * 256 + tid] = e0;
      slots[(kIn + kOut + o + 1) * 256 + tid] = e1;
    }
    bool any = false;
#pragma unroll
    for (int k = 0; k < kOut; ++k) {
      const float x = slots[(kIn + k) * 256 + tid], y = slots[(kIn + kOut + k) * 256 + tid];
      if (__builtin_bit_cast(unsigned, x) != __builtin_bit_cast(unsigned, y)) {
        any = true;
        const unsigned e = atomicAdd(p.n_examples, 1u);
        if (e < 64) {
          p.examples[e * 4 + 0] = (float)(TEST * 16 + VAR) + 0.01f * k; p.examples[e * 4 + 1] = (float)lane;
          p.examples[e * 4 + 2] = x; p.examples[e * 4 + 3] = y;
        }
      }
    }
    if (any) { bad[lane >> 4]++; bad_it++; }
    sink += t0.x + t1.y + t2.z + t3.w + l0 + l1;
  }
  unsigned* e = p.err + p.slot * 5;
  for (int q = 0; q < 4; ++q) if (bad[q]) atomicAdd(&e[q], bad[q]);
  if (bad_it) atomicAdd(&e[4], bad_it);
  p.sink[gid] = sink + acc[0] + acc[1] + acc[2] + acc[3];
  if (PARTNER && lane == 0 && atomicAdd(&done_count, 1) == 3) done_flag = 1;     // the last test wave releases the partners
}

template <int TEST, int VAR, int PARTNER = 0>
static void launch(Params& p, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  p.slot++;
  hipLaunchKernelGGL((probe<TEST, VAR, PARTNER>), dim3(PARTNER ? 256 : 512), dim3(PARTNER ? 512 : 256), 0, 0, p);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned e[5];
  hipMemcpy(e, p.err + p.slot * 5, sizeof(e), hipMemcpyDeviceToHost);
  const double n = (PARTNER ? 256.0 : 512.0) * 256 * p.iters;
  printf("test %d variant %d  %-58s mismatching lane-iterations %9u of %.3g  (lanes 0-15: %u, 16-31: %u, 32-47: %u, 48-63: %u)  %.1f ms\n",
         TEST, VAR, what, e[4], n, e[0], e[1], e[2], e[3], ms);
  fflush(stdout);
}

int main(int argc, char** argv) {
  Params p;
  p.iters = argc > 1 ? atoi(argv[1]) : 4000;
  p.slot = -1;
  hipMalloc(&p.err, 256 * 5 * sizeof(unsigned)); hipMemset(p.err, 0, 256 * 5 * sizeof(unsigned));
  hipMalloc(&p.examples, 64 * 4 * sizeof(float)); hipMemset(p.examples, 0, 64 * 4 * sizeof(float));
  hipMalloc(&p.n_examples, 4); hipMemset(p.n_examples, 0, 4);
  p.traffic_len = 1u << 22;                                       // 64 MB of float4: misses the L2
  f32x4* tr; hipMalloc(&tr, (size_t)p.traffic_len * sizeof(f32x4)); hipMemset(tr, 0, (size_t)p.traffic_len * sizeof(f32x4));
  p.traffic = tr;
  hipMalloc(&p.sink, 512 * 512 * sizeof(float));
  printf("pk_hazard: %d iterations per lane, 512 blocks x 4 waves (2 waves / SIMD), tight vs s_nop-padded execution of the same instructions\n", p.iters);
  launch<0, 0>(p, "replica of the compiled block, as compiled");
  launch<0, 1>(p, "replica, s_nop 7 before the DPP moves");
  launch<0, 2>(p, "replica, s_nop 7 after the DPP moves");
  launch<0, 3>(p, "replica, s_nop 7 between differences and op_sel products");
  launch<0, 4>(p, "replica, s_nop 7 after the op_sel products");
  launch<0, 5>(p, "replica, s_nop 0 between differences and op_sel products");
  launch<0, 6>(p, "replica, s_nop 1 between differences and op_sel products");
  launch<0, 7>(p, "replica, s_nop 0 between all packed instructions");
  launch<1, 0>(p, "v_pk_add -> 0 wait states -> DPP read (violates the ISA rule)");
  launch<1, 1>(p, "v_pk_add -> 1 wait state  -> DPP read (violates the ISA rule)");
  launch<1, 2>(p, "v_pk_add -> 2 wait states -> DPP read (what LLVM inserts)");
  launch<1, 3>(p, "v_pk_add -> 3 wait states -> DPP read");
  launch<1, 4>(p, "v_pk_add -> 4 wait states -> DPP read");
  launch<2, 0>(p, "v_add    -> 0 wait states -> DPP read (violates the ISA rule)");
  launch<2, 1>(p, "v_add    -> 1 wait state  -> DPP read (violates the ISA rule)");
  launch<2, 2>(p, "v_add    -> 2 wait states -> DPP read (what LLVM inserts)");
  launch<2, 3>(p, "v_add    -> 3 wait states -> DPP read");
  launch<3, 0>(p, "DPP mov -> v_pk_add consumer, back to back");
  launch<3, 2>(p, "DPP mov -> 2 wait states -> v_pk_add consumer");
  launch<4, 0>(p, "v_pk_add(neg) -> v_pk_mul cross-half op_sel -> v_pk_add, back to back");
  launch<4, 2>(p, "the same with 2 wait states between them");
  printf("the replica next to a SIMD partner wave that streams one instruction class (256 blocks x 8 waves, partner = wave + 4):\n");
  launch<0, 0, 1>(p, "replica | partner: v_mfma_f32_16x16x32_f16");
  launch<0, 0, 2>(p, "replica | partner: v_mfma_f32_16x16x4_f32");
  launch<0, 0, 3>(p, "replica | partner: v_permlane32_swap / v_permlane16_swap");
  launch<0, 0, 4>(p, "replica | partner: v_exp_f32 / v_log_f32");
  launch<0, 0, 5>(p, "replica | partner: DPP row shifts");
  launch<0, 0, 6>(p, "replica | partner: ds_read_b128 / ds_write_b128");
  launch<0, 0, 7>(p, "replica | partner: global loads");
  launch<0, 0, 8>(p, "replica | partner: v_pk_fma_f32");
  launch<0, 0, 9>(p, "replica | partner: v_mfma_f32_16x16x16_f16");
  launch<0, 0, 10>(p, "replica | partner: v_mfma_f32_16x16x32_bf16");
  launch<4, 0, 1>(p, "v_pk_add(neg) -> v_pk_mul cross op_sel -> v_pk_add | partner: v_mfma_f32_16x16x32_f16");
  launch<1, 2, 1>(p, "v_pk_add -> 2 wait states -> DPP | partner: v_mfma_f32_16x16x32_f16");
  printf("single instructions between s_nop 7 pads against the compiler's scalar arithmetic, partner wave streaming v_mfma_f32_16x16x32_f16:\n");
  launch<5, 0, 1>(p, "v_pk_mul_f32 d, a, b");
  launch<5, 1, 1>(p, "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]          (src1 halves swapped)");
  launch<5, 23, 1>(p, "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0] neg on src1");
  launch<5, 2, 1>(p, "v_pk_mul_f32 d, a, b op_sel:[1,0] op_sel_hi:[0,1]          (src0 halves swapped)");
  launch<5, 13, 1>(p, "v_pk_mul_f32 d, a, b op_sel:[1,1] op_sel_hi:[0,0]          (both swapped)");
  launch<5, 3, 1>(p, "v_pk_mul_f32 d, a, b op_sel_hi:[0,1]                        (src0.lo to both)");
  launch<5, 4, 1>(p, "v_pk_mul_f32 d, a, b op_sel_hi:[1,0]                        (src1.lo to both)");
  launch<5, 14, 1>(p, "v_pk_mul_f32 d, a, b op_sel:[0,1]                           (src1.hi to both)");
  launch<5, 5, 1>(p, "v_pk_mul_f32 a, a, b op_sel:[0,1] op_sel_hi:[1,0]          (in place)");
  launch<5, 6, 1>(p, "v_pk_add_f32 d, a, b");
  launch<5, 7, 1>(p, "v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]          (src1 swapped)");
  launch<5, 22, 1>(p, "v_pk_add_f32 d, a, a op_sel:[0,1] op_sel_hi:[1,0]          (src1 swapped, same register pair)");
  launch<5, 20, 1>(p, "v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[0,0]          (src0.lo to both, src1 swapped)");
  launch<5, 21, 1>(p, "v_pk_add_f32 d, a, b op_sel:[1,1] op_sel_hi:[1,0]          (src0.hi to both, src1 swapped)");
  launch<5, 8, 1>(p, "v_pk_fma_f32 d, a, b, c");
  launch<5, 9, 1>(p, "v_pk_fma_f32 d, a, b, c op_sel:[1,0,0]                     (src0.hi to both)");
  launch<5, 10, 1>(p, "v_pk_fma_f32 d, a, b, c op_sel_hi:[0,1,1]                  (src0.lo to both)");
  launch<5, 17, 1>(p, "v_pk_fma_f32 d, a, b, c op_sel:[1,0,0] op_sel_hi:[0,1,1]   (src0 swapped)");
  launch<5, 15, 1>(p, "v_pk_fma_f32 d, a, b, c op_sel:[0,1,0] op_sel_hi:[1,0,1]   (src1 swapped)");
  launch<5, 16, 1>(p, "v_pk_fma_f32 d, a, b, c op_sel:[0,0,1] op_sel_hi:[1,1,0]   (src2 swapped)");
  launch<5, 18, 1>(p, "v_pk_mov_b32 d, a, b op_sel:[0,1]   (expect a.lo, b.hi)");
  launch<5, 19, 1>(p, "v_pk_mov_b32 d, a, b op_sel:[1,0]   (expect a.hi, b.lo)");
  launch<5, 11, 1>(p, "v_mul_f32 x 2 (control)");
  launch<5, 12, 1>(p, "v_fma_f32 x 2 (control)");
  printf("the same without a partner stream (2 waves / SIMD of the probe itself):\n");
  launch<5, 1, 0>(p, "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]");
  launch<5, 7, 0>(p, "v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]");
  printf("src1-swapped v_pk_mul_f32 next to other matrix instructions:\n");
  launch<5, 1, 2>(p, "partner v_mfma_f32_16x16x4_f32");
  launch<5, 1, 10>(p, "partner v_mfma_f32_16x16x32_bf16");
  launch<5, 1, 9>(p, "partner v_mfma_f32_16x16x16_f16");
  unsigned ne; hipMemcpy(&ne, p.n_examples, 4, hipMemcpyDeviceToHost);
  float ex[64 * 4]; hipMemcpy(ex, p.examples, sizeof(ex), hipMemcpyDeviceToHost);
  for (unsigned i = 0; i < (ne < 24 ? ne : 24); ++i)
    printf("example: test/variant.output %.2f lane %2d tight %.9g padded %.9g\n", ex[i * 4], (int)ex[i * 4 + 1], ex[i * 4 + 2], ex[i * 4 + 3]);
  return 0;
}
