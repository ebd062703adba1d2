// One-off measurement quoted in DESIGN.md: the issue cost of the wave64 VALU instructions the encode kernels are made of, on gfx950.
// Every kernel is a long unrolled run of ONE instruction on eight independent registers; 1, 2 and 8 wavefronts per SIMD (256 CUs x 4 SIMDs).
// Reported: chip-wide wave-instructions per second and shader cycles per wave-instruction per SIMD at the nominal clock.
// build + run on a GPU box:  hipcc -O2 --offload-arch=gfx950 tools/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define DEFK(name, ASM)                                                                                                          \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters) {                                                      \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;    \
        uint32_t b = blockIdx.x | 1u;                                                                                            \
        uint64_t w = ((uint64_t)b << 32) | a0;                                                                                   \
        for (int i = 0; i < iters; ++i) {                                                                                        \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                      \
                asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(w)        \
                             : "v"(b), "s"(i)                                                                                    \
                             : "vcc", "s40", "s41", "s42", "s43");                                                               \
            }                                                                                                                    \
        }                                                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)w ^ (uint32_t)(w >> 32);         \
    }
// %0..%7 the eight registers, %8 a 64-bit register pair, %9 a vector operand, %10 a scalar operand
#define X8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
DEFK(k_add,      "v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n")
DEFK(k_xor,      "v_xor_b32 %0, %0, %9\n v_xor_b32 %1, %1, %9\n v_xor_b32 %2, %2, %9\n v_xor_b32 %3, %3, %9\n v_xor_b32 %4, %4, %9\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %9\n v_xor_b32 %7, %7, %9\n")
DEFK(k_lshr,     "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7\n")
DEFK(k_andor,    "v_and_or_b32 %0, %0, %9, %1\n v_and_or_b32 %1, %1, %9, %2\n v_and_or_b32 %2, %2, %9, %3\n v_and_or_b32 %3, %3, %9, %4\n v_and_or_b32 %4, %4, %9, %5\n v_and_or_b32 %5, %5, %9, %6\n v_and_or_b32 %6, %6, %9, %7\n v_and_or_b32 %7, %7, %9, %0\n")
DEFK(k_add3,     "v_add3_u32 %0, %0, %9, %1\n v_add3_u32 %1, %1, %9, %2\n v_add3_u32 %2, %2, %9, %3\n v_add3_u32 %3, %3, %9, %4\n v_add3_u32 %4, %4, %9, %5\n v_add3_u32 %5, %5, %9, %6\n v_add3_u32 %6, %6, %9, %7\n v_add3_u32 %7, %7, %9, %0\n")
DEFK(k_lshladd,  "v_lshl_add_u32 %0, %0, 2, %9\n v_lshl_add_u32 %1, %1, 2, %9\n v_lshl_add_u32 %2, %2, 2, %9\n v_lshl_add_u32 %3, %3, 2, %9\n v_lshl_add_u32 %4, %4, 2, %9\n v_lshl_add_u32 %5, %5, 2, %9\n v_lshl_add_u32 %6, %6, 2, %9\n v_lshl_add_u32 %7, %7, 2, %9\n")
DEFK(k_alignbit, "v_alignbit_b32 %0, %0, %9, 7\n v_alignbit_b32 %1, %1, %9, 7\n v_alignbit_b32 %2, %2, %9, 7\n v_alignbit_b32 %3, %3, %9, 7\n v_alignbit_b32 %4, %4, %9, 7\n v_alignbit_b32 %5, %5, %9, 7\n v_alignbit_b32 %6, %6, %9, 7\n v_alignbit_b32 %7, %7, %9, 7\n")
DEFK(k_alignbitv,"v_alignbit_b32 %0, %0, %9, %1\n v_alignbit_b32 %1, %1, %9, %2\n v_alignbit_b32 %2, %2, %9, %3\n v_alignbit_b32 %3, %3, %9, %4\n v_alignbit_b32 %4, %4, %9, %5\n v_alignbit_b32 %5, %5, %9, %6\n v_alignbit_b32 %6, %6, %9, %7\n v_alignbit_b32 %7, %7, %9, %0\n")
DEFK(k_bfe,      "v_bfe_u32 %0, %0, 1, 30\n v_bfe_u32 %1, %1, 1, 30\n v_bfe_u32 %2, %2, 1, 30\n v_bfe_u32 %3, %3, 1, 30\n v_bfe_u32 %4, %4, 1, 30\n v_bfe_u32 %5, %5, 1, 30\n v_bfe_u32 %6, %6, 1, 30\n v_bfe_u32 %7, %7, 1, 30\n")
DEFK(k_min3,     "v_min3_u32 %0, %0, %9, %1\n v_min3_u32 %1, %1, %9, %2\n v_min3_u32 %2, %2, %9, %3\n v_min3_u32 %3, %3, %9, %4\n v_min3_u32 %4, %4, %9, %5\n v_min3_u32 %5, %5, %9, %6\n v_min3_u32 %6, %6, %9, %7\n v_min3_u32 %7, %7, %9, %0\n")
DEFK(k_mullo,    "v_mul_lo_u32 %0, %0, %9\n v_mul_lo_u32 %1, %1, %9\n v_mul_lo_u32 %2, %2, %9\n v_mul_lo_u32 %3, %3, %9\n v_mul_lo_u32 %4, %4, %9\n v_mul_lo_u32 %5, %5, %9\n v_mul_lo_u32 %6, %6, %9\n v_mul_lo_u32 %7, %7, %9\n")
DEFK(k_mulhi,    "v_mul_hi_u32 %0, %0, %9\n v_mul_hi_u32 %1, %1, %9\n v_mul_hi_u32 %2, %2, %9\n v_mul_hi_u32 %3, %3, %9\n v_mul_hi_u32 %4, %4, %9\n v_mul_hi_u32 %5, %5, %9\n v_mul_hi_u32 %6, %6, %9\n v_mul_hi_u32 %7, %7, %9\n")
DEFK(k_mul24,    "v_mul_u32_u24 %0, %0, %9\n v_mul_u32_u24 %1, %1, %9\n v_mul_u32_u24 %2, %2, %9\n v_mul_u32_u24 %3, %3, %9\n v_mul_u32_u24 %4, %4, %9\n v_mul_u32_u24 %5, %5, %9\n v_mul_u32_u24 %6, %6, %9\n v_mul_u32_u24 %7, %7, %9\n")
DEFK(k_mad24,    "v_mad_u32_u24 %0, %0, %9, %1\n v_mad_u32_u24 %1, %1, %9, %2\n v_mad_u32_u24 %2, %2, %9, %3\n v_mad_u32_u24 %3, %3, %9, %4\n v_mad_u32_u24 %4, %4, %9, %5\n v_mad_u32_u24 %5, %5, %9, %6\n v_mad_u32_u24 %6, %6, %9, %7\n v_mad_u32_u24 %7, %7, %9, %0\n")
DEFK(k_cndvcc,   "v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n")
DEFK(k_cndsgpr,  "v_cndmask_b32_e64 %0, %0, %9, s[40:41]\n v_cndmask_b32_e64 %1, %1, %9, s[40:41]\n v_cndmask_b32_e64 %2, %2, %9, s[40:41]\n v_cndmask_b32_e64 %3, %3, %9, s[40:41]\n v_cndmask_b32_e64 %4, %4, %9, s[40:41]\n v_cndmask_b32_e64 %5, %5, %9, s[40:41]\n v_cndmask_b32_e64 %6, %6, %9, s[40:41]\n v_cndmask_b32_e64 %7, %7, %9, s[40:41]\n")
DEFK(k_cmpvcc,   "v_cmp_eq_u32 vcc, %0, %9\n v_cmp_eq_u32 vcc, %1, %9\n v_cmp_eq_u32 vcc, %2, %9\n v_cmp_eq_u32 vcc, %3, %9\n v_cmp_eq_u32 vcc, %4, %9\n v_cmp_eq_u32 vcc, %5, %9\n v_cmp_eq_u32 vcc, %6, %9\n v_cmp_eq_u32 vcc, %7, %9\n")
DEFK(k_cmpsgpr,  "v_cmp_eq_u32_e64 s[40:41], %0, %9\n v_cmp_eq_u32_e64 s[42:43], %1, %9\n v_cmp_eq_u32_e64 s[40:41], %2, %9\n v_cmp_eq_u32_e64 s[42:43], %3, %9\n v_cmp_eq_u32_e64 s[40:41], %4, %9\n v_cmp_eq_u32_e64 s[42:43], %5, %9\n v_cmp_eq_u32_e64 s[40:41], %6, %9\n v_cmp_eq_u32_e64 s[42:43], %7, %9\n")
DEFK(k_cmp64,    "v_cmp_eq_u64 vcc, %8, %8\n v_cmp_eq_u64 vcc, %8, %8\n v_cmp_eq_u64 vcc, %8, %8\n v_cmp_eq_u64 vcc, %8, %8\n v_cmp_eq_u64 vcc, %8, %8\n v_cmp_eq_u64 vcc, %8, %8\n v_cmp_eq_u64 vcc, %8, %8\n v_cmp_eq_u64 vcc, %8, %8\n")
DEFK(k_lshladd64,"v_lshl_add_u64 %8, %8, 2, %8\n v_lshl_add_u64 %8, %8, 2, %8\n v_lshl_add_u64 %8, %8, 2, %8\n v_lshl_add_u64 %8, %8, 2, %8\n v_lshl_add_u64 %8, %8, 2, %8\n v_lshl_add_u64 %8, %8, 2, %8\n v_lshl_add_u64 %8, %8, 2, %8\n v_lshl_add_u64 %8, %8, 2, %8\n")
DEFK(k_lshl64,   "v_lshlrev_b64 %8, 3, %8\n v_lshlrev_b64 %8, 3, %8\n v_lshlrev_b64 %8, 3, %8\n v_lshlrev_b64 %8, 3, %8\n v_lshlrev_b64 %8, 3, %8\n v_lshlrev_b64 %8, 3, %8\n v_lshlrev_b64 %8, 3, %8\n v_lshlrev_b64 %8, 3, %8\n")
DEFK(k_bcnt,     "v_bcnt_u32_b32 %0, %0, %9\n v_bcnt_u32_b32 %1, %1, %9\n v_bcnt_u32_b32 %2, %2, %9\n v_bcnt_u32_b32 %3, %3, %9\n v_bcnt_u32_b32 %4, %4, %9\n v_bcnt_u32_b32 %5, %5, %9\n v_bcnt_u32_b32 %6, %6, %9\n v_bcnt_u32_b32 %7, %7, %9\n")
DEFK(k_mbcnt,    "v_mbcnt_lo_u32_b32 %0, %0, %9\n v_mbcnt_hi_u32_b32 %1, %1, %9\n v_mbcnt_lo_u32_b32 %2, %2, %9\n v_mbcnt_hi_u32_b32 %3, %3, %9\n v_mbcnt_lo_u32_b32 %4, %4, %9\n v_mbcnt_hi_u32_b32 %5, %5, %9\n v_mbcnt_lo_u32_b32 %6, %6, %9\n v_mbcnt_hi_u32_b32 %7, %7, %9\n")
DEFK(k_ffbl,     "v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7\n")
DEFK(k_perm,     "v_perm_b32 %0, %0, %9, %1\n v_perm_b32 %1, %1, %9, %2\n v_perm_b32 %2, %2, %9, %3\n v_perm_b32 %3, %3, %9, %4\n v_perm_b32 %4, %4, %9, %5\n v_perm_b32 %5, %5, %9, %6\n v_perm_b32 %6, %6, %9, %7\n v_perm_b32 %7, %7, %9, %0\n")
DEFK(k_dpp,      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
DEFK(k_sdwa,     "v_xor_b32_sdwa %0, %0, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %2, %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %3, %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %4, %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %6, %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
DEFK(k_readlane, "v_readlane_b32 s40, %0, 3\n v_readlane_b32 s41, %1, 3\n v_readlane_b32 s40, %2, 3\n v_readlane_b32 s41, %3, 3\n v_readlane_b32 s40, %4, 3\n v_readlane_b32 s41, %5, 3\n v_readlane_b32 s40, %6, 3\n v_readlane_b32 s41, %7, 3\n")
DEFK(k_xorsgpr,  "v_xor_b32 %0, %10, %0\n v_xor_b32 %1, %10, %1\n v_xor_b32 %2, %10, %2\n v_xor_b32 %3, %10, %3\n v_xor_b32 %4, %10, %4\n v_xor_b32 %5, %10, %5\n v_xor_b32 %6, %10, %6\n v_xor_b32 %7, %10, %7\n")
DEFK(k_fma,      "v_fma_f32 %0, %0, %9, %0\n v_fma_f32 %1, %1, %9, %1\n v_fma_f32 %2, %2, %9, %2\n v_fma_f32 %3, %3, %9, %3\n v_fma_f32 %4, %4, %9, %4\n v_fma_f32 %5, %5, %9, %5\n v_fma_f32 %6, %6, %9, %6\n v_fma_f32 %7, %7, %9, %7\n")
DEFK(k_mix_salu, "v_add_u32 %0, %0, %9\n s_add_u32 s40, s40, 1\n v_add_u32 %1, %1, %9\n s_and_b32 s41, s41, s40\n v_add_u32 %2, %2, %9\n s_add_u32 s42, s42, 1\n v_add_u32 %3, %3, %9\n s_and_b32 s43, s43, s42\n v_add_u32 %4, %4, %9\n s_add_u32 s40, s40, 1\n v_add_u32 %5, %5, %9\n s_and_b32 s41, s41, s40\n v_add_u32 %6, %6, %9\n s_add_u32 s42, s42, 1\n v_add_u32 %7, %7, %9\n s_and_b32 s43, s43, s42\n")

DEFK(k_add_e64, "v_add_u32_e64 %0, %0, %9\n v_add_u32_e64 %1, %1, %9\n v_add_u32_e64 %2, %2, %9\n v_add_u32_e64 %3, %3, %9\n v_add_u32_e64 %4, %4, %9\n v_add_u32_e64 %5, %5, %9\n v_add_u32_e64 %6, %6, %9\n v_add_u32_e64 %7, %7, %9\n ")
DEFK(k_xor_e64, "v_xor_b32_e64 %0, %0, %9\n v_xor_b32_e64 %1, %1, %9\n v_xor_b32_e64 %2, %2, %9\n v_xor_b32_e64 %3, %3, %9\n v_xor_b32_e64 %4, %4, %9\n v_xor_b32_e64 %5, %5, %9\n v_xor_b32_e64 %6, %6, %9\n v_xor_b32_e64 %7, %7, %9\n ")
DEFK(k_mov, "v_mov_b32 %0, %9\n v_mov_b32 %1, %9\n v_mov_b32 %2, %9\n v_mov_b32 %3, %9\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n ")
DEFK(k_not, "v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3\n v_not_b32 %4, %4\n v_not_b32 %5, %5\n v_not_b32 %6, %6\n v_not_b32 %7, %7\n ")
DEFK(k_xorlit, "v_xor_b32 %0, 0x12345678, %0\n v_xor_b32 %1, 0x12345678, %1\n v_xor_b32 %2, 0x12345678, %2\n v_xor_b32 %3, 0x12345678, %3\n v_xor_b32 %4, 0x12345678, %4\n v_xor_b32 %5, 0x12345678, %5\n v_xor_b32 %6, 0x12345678, %6\n v_xor_b32 %7, 0x12345678, %7\n ")
DEFK(k_xorinl, "v_xor_b32 %0, 7, %0\n v_xor_b32 %1, 7, %1\n v_xor_b32 %2, 7, %2\n v_xor_b32 %3, 7, %3\n v_xor_b32 %4, 7, %4\n v_xor_b32 %5, 7, %5\n v_xor_b32 %6, 7, %6\n v_xor_b32 %7, 7, %7\n ")
DEFK(k_addc, "v_add_co_u32 %0, vcc, %0, %9\n v_add_co_u32 %1, vcc, %1, %9\n v_add_co_u32 %2, vcc, %2, %9\n v_add_co_u32 %3, vcc, %3, %9\n v_add_co_u32 %4, vcc, %4, %9\n v_add_co_u32 %5, vcc, %5, %9\n v_add_co_u32 %6, vcc, %6, %9\n v_add_co_u32 %7, vcc, %7, %9\n ")
DEFK(k_cmpcnd, "v_cmp_lt_u32 vcc, %0, %9\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_u32 vcc, %1, %9\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_u32 vcc, %2, %9\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_u32 vcc, %3, %9\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_lt_u32 vcc, %4, %9\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_lt_u32 vcc, %5, %9\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_u32 vcc, %6, %9\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_lt_u32 vcc, %7, %9\n v_cndmask_b32 %7, %7, %9, vcc\n ")
DEFK(k_cmpcnd64, "v_cmp_lt_u32_e64 s[40:41], %0, %9\n v_cndmask_b32_e64 %0, %0, %9, s[40:41]\n v_cmp_lt_u32_e64 s[40:41], %1, %9\n v_cndmask_b32_e64 %1, %1, %9, s[40:41]\n v_cmp_lt_u32_e64 s[40:41], %2, %9\n v_cndmask_b32_e64 %2, %2, %9, s[40:41]\n v_cmp_lt_u32_e64 s[40:41], %3, %9\n v_cndmask_b32_e64 %3, %3, %9, s[40:41]\n v_cmp_lt_u32_e64 s[40:41], %4, %9\n v_cndmask_b32_e64 %4, %4, %9, s[40:41]\n v_cmp_lt_u32_e64 s[40:41], %5, %9\n v_cndmask_b32_e64 %5, %5, %9, s[40:41]\n v_cmp_lt_u32_e64 s[40:41], %6, %9\n v_cndmask_b32_e64 %6, %6, %9, s[40:41]\n v_cmp_lt_u32_e64 s[40:41], %7, %9\n v_cndmask_b32_e64 %7, %7, %9, s[40:41]\n ")
DEFK(k_fma2, "v_fma_f32 %0, %0, %9, %1\n v_fma_f32 %1, %1, %9, %2\n v_fma_f32 %2, %2, %9, %3\n v_fma_f32 %3, %3, %9, %4\n v_fma_f32 %4, %4, %9, %5\n v_fma_f32 %5, %5, %9, %6\n v_fma_f32 %6, %6, %9, %7\n v_fma_f32 %7, %7, %9, %0\n ")
DEFK(k_or3, "v_or3_b32 %0, %0, %9, %1\n v_or3_b32 %1, %1, %9, %2\n v_or3_b32 %2, %2, %9, %3\n v_or3_b32 %3, %3, %9, %4\n v_or3_b32 %4, %4, %9, %5\n v_or3_b32 %5, %5, %9, %6\n v_or3_b32 %6, %6, %9, %7\n v_or3_b32 %7, %7, %9, %0\n ")
DEFK(k_xad, "v_xad_u32 %0, %0, %9, %1\n v_xad_u32 %1, %1, %9, %2\n v_xad_u32 %2, %2, %9, %3\n v_xad_u32 %3, %3, %9, %4\n v_xad_u32 %4, %4, %9, %5\n v_xad_u32 %5, %5, %9, %6\n v_xad_u32 %6, %6, %9, %7\n v_xad_u32 %7, %7, %9, %0\n ")
DEFK(k_pkadd, "v_pk_add_u16 %0, %0, %9\n v_pk_add_u16 %1, %1, %9\n v_pk_add_u16 %2, %2, %9\n v_pk_add_u16 %3, %3, %9\n v_pk_add_u16 %4, %4, %9\n v_pk_add_u16 %5, %5, %9\n v_pk_add_u16 %6, %6, %9\n v_pk_add_u16 %7, %7, %9\n ")
DEFK(k_max, "v_max_u32 %0, %0, %9\n v_max_u32 %1, %1, %9\n v_max_u32 %2, %2, %9\n v_max_u32 %3, %3, %9\n v_max_u32 %4, %4, %9\n v_max_u32 %5, %5, %9\n v_max_u32 %6, %6, %9\n v_max_u32 %7, %7, %9\n ")
DEFK(k_sub, "v_sub_u32 %0, %0, %9\n v_sub_u32 %1, %1, %9\n v_sub_u32 %2, %2, %9\n v_sub_u32 %3, %3, %9\n v_sub_u32 %4, %4, %9\n v_sub_u32 %5, %5, %9\n v_sub_u32 %6, %6, %9\n v_sub_u32 %7, %7, %9\n ")
DEFK(k_and, "v_and_b32 %0, %0, %9\n v_and_b32 %1, %1, %9\n v_and_b32 %2, %2, %9\n v_and_b32 %3, %3, %9\n v_and_b32 %4, %4, %9\n v_and_b32 %5, %5, %9\n v_and_b32 %6, %6, %9\n v_and_b32 %7, %7, %9\n ")
DEFK(k_lshlv, "v_lshlrev_b32 %0, %9, %0\n v_lshlrev_b32 %1, %9, %1\n v_lshlrev_b32 %2, %9, %2\n v_lshlrev_b32 %3, %9, %3\n v_lshlrev_b32 %4, %9, %4\n v_lshlrev_b32 %5, %9, %5\n v_lshlrev_b32 %6, %9, %6\n v_lshlrev_b32 %7, %9, %7\n ")
typedef void (*kern_t)(uint32_t*, int);
static void run(const char* name, kern_t k, int waves_per_simd, int per_iter) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;               // 256 threads = one wavefront per SIMD; waves_per_simd blocks per CU
    const int iters = 2048;
    uint32_t* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * 8 * per_iter;
    const double rate = instr / (ms * 1e-3);
    printf("%-12s waves/SIMD %d: %7.1f G wave-instr/s chip-wide = %.2f cycles per wave-instruction per SIMD at %d MHz\n", name, waves_per_simd, rate / 1e9,
           (double)prop.multiProcessorCount * 4 * prop.clockRate * 1e3 / rate, prop.clockRate / 1000);
    (void)hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 8}) run("v_add_u32", k_add, w, 8);
#define RUN(n) run(#n, n, 8, 8);
    RUN(k_xor) RUN(k_lshr) RUN(k_andor) RUN(k_add3) RUN(k_lshladd) RUN(k_alignbit) RUN(k_alignbitv) RUN(k_bfe) RUN(k_min3) RUN(k_mullo) RUN(k_mulhi) RUN(k_mul24) RUN(k_mad24)
    RUN(k_cndvcc) RUN(k_cndsgpr) RUN(k_cmpvcc) RUN(k_cmpsgpr) RUN(k_cmp64) RUN(k_lshladd64) RUN(k_lshl64) RUN(k_bcnt) RUN(k_mbcnt) RUN(k_ffbl) RUN(k_perm) RUN(k_dpp) RUN(k_sdwa)
    RUN(k_readlane) RUN(k_xorsgpr) RUN(k_fma)
    RUN(k_add_e64) RUN(k_xor_e64) RUN(k_mov) RUN(k_not) RUN(k_xorlit) RUN(k_xorinl) RUN(k_addc) RUN(k_fma2) RUN(k_or3) RUN(k_xad) RUN(k_pkadd) RUN(k_max) RUN(k_sub) RUN(k_and) RUN(k_lshlv)
    run("cmp+cnd vcc", k_cmpcnd, 8, 16); run("cmp+cnd sgpr", k_cmpcnd64, 8, 16);
    RUN(k_xor) RUN(k_ffbl) RUN(k_cmpvcc)
    run("v_add+s_add", k_mix_salu, 8, 16);
    return 0;
}
