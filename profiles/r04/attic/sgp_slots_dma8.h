// ---- the paired kernel's half 0: the slot sequence WITH the wave's share of the copy
// of the next stage's A chunk in it.  A 1 KB global_load_lds takes the issuing wave
// 64 .. 290 cycles; sixteen of them in one burst in front of the slots hold the half
// back by a thousand cycles while its partner half multiplies alone (config 3:
// 15.3 ms bunched in front, 14.7 behind, 14.2 spread -- profiles/r04/experiments.txt).
// So one group (2 KB: the wave's next global slot, 4 slots on) goes out behind every
// even position that runs, and what a thin stage has not placed follows the sequence.
// Running state in hand-named scalar registers: s95 groups to go, s[96:97] source,
// s94 LDS address (M0); s92 / s93 scratch.
// ctl = groups | active slots << 8 | first << 16 | skip0 << 17.
#define SGP_DMA_GROUP                                                        \
  "s_mov_b32 m0, s94\n\t"                                                    \
  "s_sub_u32 s95, s95, 1\n\t"                                                \
  "global_load_lds_dwordx4 %[voff], s[96:97]\n\t"                            \
  "global_load_lds_dwordx4 %[voff], s[96:97] offset:1024\n\t"                \
  "s_sub_u32 s96, s96, %[step]\n\t"                                          \
  "s_subb_u32 s97, s97, 0\n\t"                                               \
  "s_add_u32 s94, s94, 8192\n\t"
#define SGP_DMA_HOOK(V, P)                                                   \
  "s_cmp_eq_u32 s95, 0\n\t"                                                  \
  "s_cbranch_scc1 .Lsgp_h" #V #P "_%=\n\t" SGP_DMA_GROUP ".Lsgp_h" #V #P "_%=:\n\t"
// even positions carry a hook
#define SGP_POS_E(OFS, V, M16, P, S, CUR, NXT)                                                   \
  ".Lsgp_" #V #P "_%=:\n\t" SGP_READ_SLOT(OFS, (S) - 1, NXT) M16(S, CUR) SGP_DMA_HOOK(V, P)      \
  "s_waitcnt lgkmcnt(0)\n\t"
#define SGP_POS_LAST_D(V, M16, P, CUR)                                                           \
  ".Lsgp_" #V #P "_%=:\n\ts_bitcmp1_b32 %[ctl], 17\n\ts_cbranch_scc1 .Lsgp_tail_%=\n\t" M16(0, CUR)
#define SGP_SEQUENCE_D(OFS, V, M16)                                                                   \
  SGP_POS_E(OFS, V, M16, 0, 15, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 1, 14, SGP_SET_B, SGP_SET_A)   \
  SGP_POS_E(OFS, V, M16, 2, 13, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 3, 12, SGP_SET_B, SGP_SET_A)   \
  SGP_POS_E(OFS, V, M16, 4, 11, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 5, 10, SGP_SET_B, SGP_SET_A)   \
  SGP_POS_E(OFS, V, M16, 6, 9, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 7, 8, SGP_SET_B, SGP_SET_A)     \
  SGP_POS_E(OFS, V, M16, 8, 7, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 9, 6, SGP_SET_B, SGP_SET_A)     \
  SGP_POS_E(OFS, V, M16, 10, 5, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 11, 4, SGP_SET_B, SGP_SET_A)   \
  SGP_POS_E(OFS, V, M16, 12, 3, SGP_SET_A, SGP_SET_B) SGP_POS(OFS, V, M16, 13, 2, SGP_SET_B, SGP_SET_A)   \
  SGP_POS_E(OFS, V, M16, 14, 1, SGP_SET_A, SGP_SET_B) SGP_POS_LAST_D(V, M16, 15, SGP_SET_B)

// nact in 1 .. 16 (2 .. 16 with skip0); groups in 0 .. 8: 2 KB pieces of the copy,
// piece i from src0 - i step to LDS address dst0 + 8192 i.
__device__ __forceinline__ void sgp_slots_dma8(int nact, int first, int skip0, unsigned abase,
                                               const double (&kb)[4][4], const SgpEntryOps& e,
                                               int groups, unsigned long long src0,
                                               unsigned step, unsigned dst0, unsigned voff) {
  unsigned ctl = unsigned(groups) | (unsigned(nact) << 8) | (first ? 1u << 16 : 0u) |
                 (skip0 ? 1u << 17 : 0u);
  ctl = __builtin_amdgcn_readfirstlane(ctl);
  step = __builtin_amdgcn_readfirstlane(step);
  dst0 = __builtin_amdgcn_readfirstlane(dst0);
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"          /* the entry operands (sgp_slots_impl8_prefetch) */
      "s_bfe_u32 s93, %[ctl], 0x80008\n\t"      /* active slots */
      "s_bitcmp1_b32 s93, 0\n\t"
      "s_cbranch_scc1 .Lsgp_odd_%=\n\t"
      "v_mov_b64 v[" SGP_STR(SGP_SET_A) ":" SGP_STR(SGP_SET_A) "+1], %[e0]\n\t"
      "v_mov_b64 v[" SGP_STR(SGP_SET_A) "+2:" SGP_STR(SGP_SET_A) "+3], %[e1]\n\t"
      "v_mov_b64 v[" SGP_STR(SGP_SET_A) "+4:" SGP_STR(SGP_SET_A) "+5], %[e2]\n\t"
      "v_mov_b64 v[" SGP_STR(SGP_SET_A) "+6:" SGP_STR(SGP_SET_A) "+7], %[e3]\n\t"
      "s_branch .Lsgp_go_%=\n\t"
      ".Lsgp_odd_%=:\n\t"
      "v_mov_b64 v[" SGP_STR(SGP_SET_B) ":" SGP_STR(SGP_SET_B) "+1], %[e0]\n\t"
      "v_mov_b64 v[" SGP_STR(SGP_SET_B) "+2:" SGP_STR(SGP_SET_B) "+3], %[e1]\n\t"
      "v_mov_b64 v[" SGP_STR(SGP_SET_B) "+4:" SGP_STR(SGP_SET_B) "+5], %[e2]\n\t"
      "v_mov_b64 v[" SGP_STR(SGP_SET_B) "+6:" SGP_STR(SGP_SET_B) "+7], %[e3]\n\t"
      ".Lsgp_go_%=:\n\t"
      "s_and_b32 s95, %[ctl], 0xff\n\t"         /* groups to go */
      "s_mov_b64 s[96:97], %[src0]\n\t"
      "s_mov_b32 s94, %[dst0]\n\t"
      /* entry = position 16 - nact: m = nact - 1 positions in front of the last one, */
      /* ceil(m / 2) of them even (with a hook), floor(m / 2) odd                      */
      "s_sub_u32 s93, s93, 1\n\t"
      "s_lshr_b32 s92, s93, 1\n\t"
      "s_mul_i32 s92, s92, .Lsgp_p15_%=-.Lsgp_p13_%=\n\t"
      "s_and_b32 s93, s93, 1\n\t"
      "s_mul_i32 s93, s93, .Lsgp_p15_%=-.Lsgp_p14_%=\n\t"
      "s_add_u32 s92, s92, s93\n\t"
      "s_getpc_b64 s[98:99]\n\t"
      ".Lsgp_base_%=:\n\t"
      "s_sub_u32 s92, .Lsgp_p15_%=-.Lsgp_base_%=, s92\n\t"
      "s_bitcmp1_b32 %[ctl], 16\n\t"
      "s_cselect_b32 s93, .Lsgp_f15_%=-.Lsgp_p15_%=, 0\n\t"
      "s_add_u32 s92, s92, s93\n\t"
      "s_add_u32 s98, s98, s92\n\t"
      "s_addc_u32 s99, s99, 0\n\t"
      "s_setpc_b64 s[98:99]\n\t"
      SGP_SEQUENCE_D(8, p, SGP_MFMA16)
      "s_branch .Lsgp_tail_%=\n\t"
      SGP_SEQUENCE_D(8, f, SGP_MFMA16_FIRST)
      /* what the stage has not placed */
      ".Lsgp_tail_%=:\n\t"
      "s_cmp_eq_u32 s95, 0\n\t"
      "s_cbranch_scc1 .Lsgp_end_%=\n\t"
      SGP_DMA_GROUP
      "s_branch .Lsgp_tail_%=\n\t"
      ".Lsgp_end_%=:"
      :
      : [ctl] "s"(ctl), [abase] "v"(abase),
        [e0] "v"(e.a[0]), [e1] "v"(e.a[1]), [e2] "v"(e.a[2]), [e3] "v"(e.a[3]),
        [b00] "v"(kb[0][0]), [b01] "v"(kb[0][1]), [b02] "v"(kb[0][2]), [b03] "v"(kb[0][3]),
        [b10] "v"(kb[1][0]), [b11] "v"(kb[1][1]), [b12] "v"(kb[1][2]), [b13] "v"(kb[1][3]),
        [b20] "v"(kb[2][0]), [b21] "v"(kb[2][1]), [b22] "v"(kb[2][2]), [b23] "v"(kb[2][3]),
        [b30] "v"(kb[3][0]), [b31] "v"(kb[3][1]), [b32] "v"(kb[3][2]), [b33] "v"(kb[3][3]),
        [src0] "s"(src0), [step] "s"(step), [dst0] "s"(dst0), [voff] "v"(voff)
      : "scc", "memory", "m0", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
        SGP_CLOBBER_SETS, SGP_CLOBBER_ACC);
}
