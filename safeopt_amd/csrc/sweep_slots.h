// The matrix phase of the 4-wave posterior sweep (sweep.hip) as ONE hand-written
// instruction sequence with hand-assigned accumulator registers.
//
// Why.  Two waves share a SIMD.  While one streams fp64 MFMAs (16 cycles each) the
// other gets ONE issue slot per MFMA, for an instruction of any kind: a stage costs
//     16.25 x MFMAs + ~6 x max(0, other instructions - MFMAs)   cycles per wave
// (profiles/r04/issue_model.txt: the measured 2.9 k cycles per wave and stage of
// config 2 and the 2.8 k of config 3 both follow from the instruction counts).  What
// is left to gain is the NUMBER of non-MFMA instructions.  The compiler-generated slot
// sequence spent a compare + branch on every slot (the active slots of a stage are
// a prefix), and every attempt to enter the sequence with one computed jump instead
// (switch with fall-through) made the register allocator copy all 64 accumulators
// at the 16 join points.  Here the accumulators are not compiler values at all:
//
//   * acc[S][m] (slot S = 16 rows of L^-1, m = point quad) lives in the ACCUMULATION
//     registers a[8 S + 2 m : 8 S + 2 m + 1], a0..a127 of the unified register file
//     (gfx90a+: MFMA takes C / D and also A / B from there).  The compiler never
//     allocates an AccVGPR on its own account; the clobbers below tell it that 128
//     of them are in use, which caps its architectural VGPRs at 128 for two waves
//     per SIMD (an attribute that reserves architectural VGPRs does not exist:
//     amdgpu_num_vgpr is not honoured);
//   * the sequence runs from slot 15 DOWN to slot 0 and is entered at slot nact - 1
//     by ONE s_setpc_b64 (positions have equal size: the assembler computes it);
//   * the first stage of an accumulator chunk (j-block 0: every slot of the chunk is
//     active) runs a second copy of the sequence whose first k-step has the
//     constant 0 as addend: the accumulators are never cleared;
//   * at the end of a chunk the squares are summed BY the matrix unit: with an
//     accumulator register as A and B operand alike one instruction forms
//     out[blk][i'][j'] = sum_i acc[blk][i][i'] acc[blk][i][j'], whose diagonal
//     i' = j' is the sum over the four rows of row group blk of the squares for
//     point j' -- no AccVGPR is ever read by the VALU;
//   * A operands of the running slot sit in one of two hand-named register sets
//     (v[112:119], v[120:127]: clobbered, so the compiler keeps them free here) while
//     the next lower slot's are read into the other (ds_read2st64_b64 with immediate
//     offsets from one base address); position p uses set p & 1.  (AccVGPRs for these
//     too would make 144: the compiler splits a wave's 256 registers 128 / 128 once
//     any AccVGPR is in use and cannot be told otherwise from the source.)  The
//     operands of the ENTRY slot are requested at the top of the stage (*_prefetch:
//     ordinary loads, ordinary compiler values -- as outputs of an asm statement the
//     compiler took them for complete and spilled them before they had arrived) and
//     the slot sequence moves them into their set: their LDS latency passes under the covariance
//     evaluation -- the two waves of a SIMD of the paired kernel reach this point
//     together, nobody would hide it for them.
//
// Hazards (nothing pads inside asm; wait states as in LLVM's GCNHazardRecognizer for
// gfx90a+ DGEMM 4x4): an accumulator is the addend again four instructions later --
// three MFMAs and the s_nop 0 between the k-step groups make the 4 wait states;
// scripts/dev/check_mfma_hazards.py scans the ISA.
//
// Operand maps of v_mfma_f64_4x4x4_4b_f64: sweep.hip ("matrix part").
#pragma once

#define SGP_STR2(x) #x
#define SGP_STR(x) SGP_STR2(x)

// accumulator of slot S, point quad M
#define SGP_ACC(S, M) "a[8*(" SGP_STR(S) ")+2*" SGP_STR(M) ":8*(" SGP_STR(S) ")+2*" SGP_STR(M) "+1]"
// k-step Q of operand set B (first register 112 or 120)
#define SGP_AOP(B, Q) "v[" SGP_STR(B) "+2*" SGP_STR(Q) ":" SGP_STR(B) "+2*" SGP_STR(Q) "+1]"

#define SGP_MFMA(S, M, B, Q) \
  "v_mfma_f64_4x4x4_4b_f64 " SGP_ACC(S, M) ", " SGP_AOP(B, Q) ", %[b" #M #Q "], " SGP_ACC(S, M) "\n\t"
// ... the first contribution to an accumulator: addend 0
#define SGP_MFMA0(S, M, B, Q) \
  "v_mfma_f64_4x4x4_4b_f64 " SGP_ACC(S, M) ", " SGP_AOP(B, Q) ", %[b" #M #Q "], 0\n\t"
// four independent accumulators per k-step; one wait state between the groups: the
// addend of an MFMA must be four wait states old
#define SGP_MFMA_Q(S, B, Q) SGP_MFMA(S, 0, B, Q) SGP_MFMA(S, 1, B, Q) SGP_MFMA(S, 2, B, Q) SGP_MFMA(S, 3, B, Q)
#define SGP_MFMA0_Q(S, B, Q) SGP_MFMA0(S, 0, B, Q) SGP_MFMA0(S, 1, B, Q) SGP_MFMA0(S, 2, B, Q) SGP_MFMA0(S, 3, B, Q)
#define SGP_QGAP "s_nop 0\n\t"
#define SGP_MFMA16(S, B) \
  SGP_MFMA_Q(S, B, 0) SGP_QGAP SGP_MFMA_Q(S, B, 1) SGP_QGAP SGP_MFMA_Q(S, B, 2) SGP_QGAP SGP_MFMA_Q(S, B, 3)
#define SGP_MFMA16_FIRST(S, B) \
  SGP_MFMA0_Q(S, B, 0) SGP_QGAP SGP_MFMA_Q(S, B, 1) SGP_QGAP SGP_MFMA_Q(S, B, 2) SGP_QGAP SGP_MFMA_Q(S, B, 3)

// read the four k-steps of slot S into set B: immediate offsets in units of 512 bytes,
// consecutive slots OFS units apart (4: the 4-wave kernel's chunk image, one slot after
// the other; 8: the paired kernel's, where a wave's slots alternate with its partner's)
#define SGP_READ_SLOT(OFS, S, B) \
  "ds_read2st64_b64 v[" SGP_STR(B) ":" SGP_STR(B) "+3], %[abase] offset0:" #OFS "*(" SGP_STR(S) ") offset1:" #OFS "*(" SGP_STR(S) ")+1\n\t" \
  "ds_read2st64_b64 v[" SGP_STR(B) "+4:" SGP_STR(B) "+7], %[abase] offset0:" #OFS "*(" SGP_STR(S) ")+2 offset1:" #OFS "*(" SGP_STR(S) ")+3\n\t"

// position of slot S: operands in set CUR, slot S - 1 read into set NXT
// (V: label prefix p = running stage, f = first stage of a chunk)
#define SGP_POS(OFS, V, M16, P, S, CUR, NXT) \
  ".Lsgp_" #V #P "_%=:\n\t" SGP_READ_SLOT(OFS, (S) - 1, NXT) M16(S, CUR) "s_waitcnt lgkmcnt(0)\n\t"
// (the last position can be skipped -- SCC set at the entry: the paired kernel's slot 0
// of half 0 may be a narrow row block, which is not one of these accumulators)
#define SGP_POS_LAST(V, M16, P, CUR) \
  ".Lsgp_" #V #P "_%=:\n\ts_cbranch_scc1 .Lsgp_end_%=\n\t" M16(0, CUR)
#define SGP_SEQUENCE(OFS, V, M16)                                                                 \
  SGP_POS(OFS, V, M16, 0, 15, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 1, 14, SGP_SET_B, SGP_SET_A)   \
  SGP_POS(OFS, V, M16, 2, 13, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 3, 12, SGP_SET_B, SGP_SET_A)   \
  SGP_POS(OFS, V, M16, 4, 11, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 5, 10, SGP_SET_B, SGP_SET_A)   \
  SGP_POS(OFS, V, M16, 6, 9, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 7, 8, SGP_SET_B, SGP_SET_A)     \
  SGP_POS(OFS, V, M16, 8, 7, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 9, 6, SGP_SET_B, SGP_SET_A)     \
  SGP_POS(OFS, V, M16, 10, 5, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 11, 4, SGP_SET_B, SGP_SET_A)   \
  SGP_POS(OFS, V, M16, 12, 3, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 13, 2, SGP_SET_B, SGP_SET_A)   \
  SGP_POS(OFS, V, M16, 14, 1, SGP_SET_A, SGP_SET_B) SGP_POS_LAST(V, M16, 15, SGP_SET_B)

#define SGP_SET_A 112
#define SGP_SET_B 120

#define SGP_CLOBBER_SETS \
  "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", \
  "v123", "v124", "v125", "v126", "v127"

// The whole accumulator file.  The compiler sizes the kernel's AccVGPR allocation by
// these names and keeps nothing of its own in them ACROSS a statement that clobbers
// them; that it touches no AccVGPR BETWEEN such statements either (as spill space or
// as an allocatable register under pressure) is checked on the generated ISA:
// scripts/dev/check_mfma_hazards.py, rule A1 -- no instruction outside these asm
// blocks may name an AccVGPR (and the file is compiled with
// -mllvm -amdgpu-spill-vgpr-to-agpr=0).
#define SGP_CLOBBER_ACC \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", \
  "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", \
  "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", \
  "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", \
  "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", \
  "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", \
  "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", \
  "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", \
  "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", \
  "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", \
  "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"

// Slots nact - 1 .. 0 of one stage (nact - 1 .. 1 with skip0: slot 0 is left out).
// abase: LDS byte address of the wave's slot 0 in the staged chunk + 8 lane;
// kb[m][q]: B operands (covariances of point quad m, k-step q, broadcast to the four
// row groups).  nact in 1 .. 16 (2 .. 16 with skip0), wave-uniform; first != 0: the
// stage is the first of its accumulator chunk.  OFS / SHIFT: slot pitch in the chunk
// image, 512 OFS = 2^SHIFT bytes.
struct SgpEntryOps {
  double a[4];      // the four k-steps of the entry slot's A operand
};
#define SGP_DEFINE_SLOTS(NAME, OFS, SHIFT)                                                          \
  /* the A operands of the entry slot nact - 1, requested early: ORDINARY loads (lds0 =  */        \
  /* the generic pointer that abase is the LDS byte address of) -- the compiler must      */        \
  /* know that they are in flight: it may copy or spill the values before the sequence   */        \
  /* waits for them                                                                        */        \
  __device__ __forceinline__ void NAME##_prefetch(int nact, const double* aT, SgpEntryOps& e) {    \
    const double* a1 = aT + (nact - 1) * (OFS * 64);                                                \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) e.a[q] = a1[q * 64];                             \
  }                                                                                                 \
  __device__ __forceinline__ void NAME(int nact, int first, int skip0, unsigned abase,             \
                                       const double (&kb)[4][4], const SgpEntryOps& e) {           \
    unsigned t0, t1;                                                                                \
    /* ("s" operands must BE scalar registers: the compiler does not always know) */                \
    nact = __builtin_amdgcn_readfirstlane(nact);                                                    \
    first = __builtin_amdgcn_readfirstlane(first);                                                  \
    skip0 = __builtin_amdgcn_readfirstlane(skip0);                                                  \
    asm volatile(                                                                                   \
        "s_waitcnt lgkmcnt(0)\n\t"          /* the entry operands (NAME##_prefetch) */             \
        /* ... into the set of their position 16 - nact */                                          \
        "s_bitcmp1_b32 %[nact], 0\n\t"                                                              \
        "s_cbranch_scc1 .Lsgp_odd_%=\n\t"                                                           \
        "v_mov_b64 v[" SGP_STR(SGP_SET_A) ":" SGP_STR(SGP_SET_A) "+1], %[e0]\n\t"                   \
        "v_mov_b64 v[" SGP_STR(SGP_SET_A) "+2:" SGP_STR(SGP_SET_A) "+3], %[e1]\n\t"                 \
        "v_mov_b64 v[" SGP_STR(SGP_SET_A) "+4:" SGP_STR(SGP_SET_A) "+5], %[e2]\n\t"                 \
        "v_mov_b64 v[" SGP_STR(SGP_SET_A) "+6:" SGP_STR(SGP_SET_A) "+7], %[e3]\n\t"                 \
        "s_branch .Lsgp_go_%=\n\t"                                                                  \
        ".Lsgp_odd_%=:\n\t"                                                                         \
        "v_mov_b64 v[" SGP_STR(SGP_SET_B) ":" SGP_STR(SGP_SET_B) "+1], %[e0]\n\t"                   \
        "v_mov_b64 v[" SGP_STR(SGP_SET_B) "+2:" SGP_STR(SGP_SET_B) "+3], %[e1]\n\t"                 \
        "v_mov_b64 v[" SGP_STR(SGP_SET_B) "+4:" SGP_STR(SGP_SET_B) "+5], %[e2]\n\t"                 \
        "v_mov_b64 v[" SGP_STR(SGP_SET_B) "+6:" SGP_STR(SGP_SET_B) "+7], %[e3]\n\t"                 \
        ".Lsgp_go_%=:\n\t"                                                                          \
        /* entry = position 15 - (nact - 1) positions of equal size, of the sequence for a     */   \
        /* running or a first stage (the two have the same layout)                              */   \
        /* (s[98:99]: the halves of a 64-bit asm operand cannot be named)                       */   \
        "s_sub_u32 %[t0], %[nact], 1\n\t"                                                           \
        "s_mul_i32 %[t1], %[t0], .Lsgp_p15_%=-.Lsgp_p14_%=\n\t"                                     \
        "s_getpc_b64 s[98:99]\n\t"                                                                  \
        ".Lsgp_base_%=:\n\t"                                                                        \
        "s_sub_u32 %[t1], .Lsgp_p15_%=-.Lsgp_base_%=, %[t1]\n\t"                                    \
        "s_cmp_lg_u32 %[first], 0\n\t"                                                              \
        "s_cselect_b32 %[t0], .Lsgp_f15_%=-.Lsgp_p15_%=, 0\n\t"                                     \
        "s_add_u32 %[t1], %[t1], %[t0]\n\t"                                                         \
        "s_add_u32 s98, s98, %[t1]\n\t"                                                             \
        "s_addc_u32 s99, s99, 0\n\t"                                                                \
        "s_cmp_lg_u32 %[skip0], 0\n\t"      /* SCC: leave out slot 0 (SGP_POS_LAST) */             \
        "s_setpc_b64 s[98:99]\n\t"         /* (the scalar work: the 2 wait states VALU -> MFMA) */  \
        SGP_SEQUENCE(OFS, p, SGP_MFMA16)                                                            \
        "s_branch .Lsgp_end_%=\n\t"                                                                 \
        SGP_SEQUENCE(OFS, f, SGP_MFMA16_FIRST)                                                      \
        ".Lsgp_end_%=:"                                                                             \
        : [t0] "=&s"(t0), [t1] "=&s"(t1)                                                            \
        : [nact] "s"(nact), [first] "s"(first), [skip0] "s"(skip0), [abase] "v"(abase),             \
          [e0] "v"(e.a[0]), [e1] "v"(e.a[1]), [e2] "v"(e.a[2]), [e3] "v"(e.a[3]),                   \
          [b00] "v"(kb[0][0]), [b01] "v"(kb[0][1]), [b02] "v"(kb[0][2]), [b03] "v"(kb[0][3]),       \
          [b10] "v"(kb[1][0]), [b11] "v"(kb[1][1]), [b12] "v"(kb[1][2]), [b13] "v"(kb[1][3]),       \
          [b20] "v"(kb[2][0]), [b21] "v"(kb[2][1]), [b22] "v"(kb[2][2]), [b23] "v"(kb[2][3]),       \
          [b30] "v"(kb[3][0]), [b31] "v"(kb[3][1]), [b32] "v"(kb[3][2]), [b33] "v"(kb[3][3])        \
        : "scc", "memory", "s98", "s99", SGP_CLOBBER_SETS, SGP_CLOBBER_ACC);                        \
  }
// the 4-wave kernel: slots 2 KB apart.  (The paired kernel on this sequence -- slots 4 KB
// apart, its LDS-DMA groups between the positions -- was built and measured: equal at
// configs 4 / 5, 1.7 % slower at config 3; profiles/r04/attic, experiments.txt.)
SGP_DEFINE_SLOTS(sgp_slots_impl4, 4, 11)
__device__ __forceinline__ void sgp_slots_prefetch(int nact, const double* aT, SgpEntryOps& e) {
  sgp_slots_impl4_prefetch(nact, aT, e);
}
__device__ __forceinline__ void sgp_slots(int nact, int first, unsigned abase,
                                          const double (&kb)[4][4], const SgpEntryOps& e) {
  sgp_slots_impl4(nact, first, 0, abase, kb, e);
}

// End of an accumulator chunk with nsl full slots (1 .. 16): sq[m] +=
// acc[S][m]^T acc[S][m] over the slots S < nsl, by the matrix unit.  Lane
// 16 i' + 4 blk + j' of sq[m] then holds sum_S sum_i acc[S][m][blk][i][i'] *
// acc[S][m][blk][i][j']: on the diagonal lanes i' = j' the squares for point 4 m + j',
// summed over the rows of row group blk and over the slots.
// (s_nop 7 twice: the last MFMA of the stage wrote an accumulator right in front of
// this; sixteen wait states cover every MFMA-result-to-MFMA-operand rule.)
#define SGP_SQ_SLOT(P, S)                                                                  \
  ".Lsgp_q" #P "_%=:\n\t"                                                                   \
  "v_mfma_f64_4x4x4_4b_f64 %[q0], " SGP_ACC(S, 0) ", " SGP_ACC(S, 0) ", %[q0]\n\t"           \
  "v_mfma_f64_4x4x4_4b_f64 %[q1], " SGP_ACC(S, 1) ", " SGP_ACC(S, 1) ", %[q1]\n\t"           \
  "v_mfma_f64_4x4x4_4b_f64 %[q2], " SGP_ACC(S, 2) ", " SGP_ACC(S, 2) ", %[q2]\n\t"           \
  "v_mfma_f64_4x4x4_4b_f64 %[q3], " SGP_ACC(S, 3) ", " SGP_ACC(S, 3) ", %[q3]\n\t" SGP_QGAP
__device__ __forceinline__ void sgp_fold_slots(int nsl, double (&sq)[4], int skip0 = 0) {
  unsigned t1;
  nsl = __builtin_amdgcn_readfirstlane(nsl);
  skip0 = __builtin_amdgcn_readfirstlane(skip0);
  asm volatile(
      "s_nop 7\n\ts_nop 7\n\t"
      "s_getpc_b64 s[98:99]\n\t"
      ".Lsgp_qbase_%=:\n\t"
      // position of slot nsl - 1 = (the last one) - (nsl - 1) positions of equal size
      "s_mul_i32 %[t1], %[nsl], .Lsgp_q14_%=-.Lsgp_q13_%=\n\t"
      "s_sub_u32 %[t1], .Lsgp_qlast_%=-.Lsgp_qbase_%=, %[t1]\n\t"
      "s_add_u32 %[t1], %[t1], .Lsgp_q14_%=-.Lsgp_q13_%=\n\t"
      "s_add_u32 s98, s98, %[t1]\n\t"
      "s_addc_u32 s99, s99, 0\n\t"
      "s_cmp_lg_u32 %[skip0], 0\n\t"        // SCC: leave out slot 0
      "s_setpc_b64 s[98:99]\n\t"
      SGP_SQ_SLOT(0, 15) SGP_SQ_SLOT(1, 14) SGP_SQ_SLOT(2, 13) SGP_SQ_SLOT(3, 12)
      SGP_SQ_SLOT(4, 11) SGP_SQ_SLOT(5, 10) SGP_SQ_SLOT(6, 9) SGP_SQ_SLOT(7, 8)
      SGP_SQ_SLOT(8, 7) SGP_SQ_SLOT(9, 6) SGP_SQ_SLOT(10, 5) SGP_SQ_SLOT(11, 4)
      SGP_SQ_SLOT(12, 3) SGP_SQ_SLOT(13, 2) SGP_SQ_SLOT(14, 1)
      ".Lsgp_qlast_%=:\n\ts_cbranch_scc1 .Lsgp_qend_%=\n\t"
      SGP_SQ_SLOT(15, 0)
      ".Lsgp_qend_%=:\n\t"
      "s_nop 7"       // (the VALU reads sq next: 6 wait states)
      : [q0] "+v"(sq[0]), [q1] "+v"(sq[1]), [q2] "+v"(sq[2]), [q3] "+v"(sq[3]), [t1] "=&s"(t1)
      : [nsl] "s"(nsl), [skip0] "s"(skip0)
      : "scc", "s98", "s99", SGP_CLOBBER_ACC);
}
