// xstep_bench.hip — what does ONE bit-step of the arithmetic-coder recurrence (arithmetic_coder.rs:41-65) cost a lone
// wavefront, in the forms the X-wave of k_coder_x3 could take?  (Round 2: the 524,288-step chain per block is the
// block-count-independent floor of the whole encoder.)
//   V0  the shipped step: state (x1, x2, d), 15 VALU, operands (p32, bitmask) 8 B/step
//   V1  state (x1 raw, d): 13 VALU, same operands; the top-bit fix-ups of x1 are left to the consumer of the tokens
//   V2  state (x1 raw, d): 11 VALU with two v_mad_u64_u32, operands (q, b01, z, z) 16 B/step
//   V3  the step k_coder_x4 ships (round 2): operands (z, z, q), one ds_read_b96 per step
//   V4  the same step, the operands of FOUR steps in three ds_read_b128 (z0 z0 z1 z1 | z2 z2 z3 z3 | q0 q1 q2 q3): 0.75 LDS reads per step
// plus issue-rate probes of a lone wave (dependent / K independent chains, VOP2 / VOP3, LDS instructions in between).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/xstep_bench tools/xstep_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------------------------
// issue-rate probes
// ------------------------------------------------------------------------------------------------------------------
#define REP 4096
struct Stamp { uint64_t t0, t1, r0, r1; };
#define PROBE(NAME, PRE, BODY)                                                                               \
    __global__ void NAME(uint32_t *out, Stamp *st, uint32_t a, uint32_t b) {                                 \
        __shared__ uint64_t lds[1024];                                                                       \
        uint32_t x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, y = b; \
        uint32_t la = threadIdx.x * 8u; (void)la; lds[threadIdx.x] = a;                                       \
        PRE;                                                                                                 \
        uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                   \
        _Pragma("unroll 8") for (int i = 0; i < REP; i++) { BODY; }                                          \
        uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                   \
        out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y + (uint32_t)lds[threadIdx.x ^ 1];       \
        if (threadIdx.x == 0 && blockIdx.x == 0) { st->t0 = t0; st->t1 = t1; st->r0 = r0; st->r1 = r1; }     \
    }

PROBE(p_dep1, , asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(y)))
PROBE(p_ind2, , asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(x0), "+v"(x1) : "v"(y)))
PROBE(p_ind4, , asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y)))
PROBE(p_ind8, , asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y)))
PROBE(p_dep1_vop3, , asm volatile("v_add3_u32 %0, %0, %1, 1" : "+v"(x0) : "v"(y)))
PROBE(p_ind4_vop3, , asm volatile("v_add3_u32 %0, %0, %4, 1\n v_add3_u32 %1, %1, %4, 1\n v_add3_u32 %2, %2, %4, 1\n v_add3_u32 %3, %3, %4, 1"
                                  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y)))
PROBE(p_dep1_prio, asm volatile("s_setprio 3"), asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(y)))
PROBE(p_dep_mulhi, , asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x0) : "v"(y)))
PROBE(p_dep_mad64, , uint64_t q = ((uint64_t)x1 << 32) | x0; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(y), "v"(x2) : "vcc"); x0 = (uint32_t)q; x1 = (uint32_t)(q >> 32))
PROBE(p_dep_add_dswrite, , asm volatile("v_add_u32 %0, %0, %1\n ds_write_b32 %2, %0" : "+v"(x0) : "v"(y), "v"(la) : "memory"))
PROBE(p_dep4_dswrite, , asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n ds_write_b32 %2, %0" : "+v"(x0) : "v"(y), "v"(la) : "memory"))
PROBE(p_dep4_dsread, , asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n ds_read_b32 %3, %2" : "+v"(x0) : "v"(y), "v"(la), "v"(x7) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)"))
PROBE(p_dep4_snop, , asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(x0) : "v"(y)))
PROBE(p_dep4_salu, , asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(x0) : "v"(y) : "s20", "scc"))

// ------------------------------------------------------------------------------------------------------------------
// step variants, one asm statement per chunk of CH input bytes (8*CH steps), operands and tokens in LDS rings
// ------------------------------------------------------------------------------------------------------------------
#define S_(x) #x
#define S(x) S_(x)

// ---- V0: shipped step (x1 %0, x2 %1, d %2), ops (p32, mask) in v[PB+2e], v[PB+2e+1], token -> v[TK], v[TK+1]
#define V0_STEP(P32, MSK, TK0, TK1)                                                  \
    "v_mul_hi_u32 v96, %2, " P32 "\n"                                                \
    "v_add_u32 v97, %0, v96\n"                                                       \
    "v_add3_u32 v96, %0, v96, 1\n"                                                   \
    "v_bfi_b32 %1, " MSK ", v97, %1\n"                                               \
    "v_bfi_b32 " TK0 ", " MSK ", %0, v96\n"                                          \
    "v_bfi_b32 v96, " TK0 ", %1, -1\n"                                               \
    "v_lshl_or_b32 v96, v96, 1, 1\n"                                                 \
    "v_bitop3_b32 v97, v96, " TK0 ", %1 bitop3:0x60\n"                               \
    "v_ffbh_u32 " TK1 ", v97\n"                                                      \
    "v_add_u32 v96, 1, %1\n"                                                         \
    "v_lshlrev_b32 %0, " TK1 ", " TK0 "\n"                                           \
    "v_lshl_add_u32 %1, v96, " TK1 ", -1\n"                                          \
    "v_sub_u32 %2, %1, %0\n"                                                         \
    "v_and_b32 %0, 0x7fffffff, %0\n"                                                 \
    "v_or_b32 %1, 0x80000000, %1\n"

// ---- V1: (x1 raw %0, d %2; %1 unused), 13 VALU
#define V1_STEP(P32, MSK, TK0, TK1)                                                  \
    "v_mul_hi_u32 v96, %2, " P32 "\n"                                                \
    "v_add3_u32 v97, %0, v96, 1\n"                                                   \
    "v_bfi_b32 " TK0 ", " MSK ", %0, v97\n"                                          \
    "v_xad_u32 v97, v96, -1, %2\n"                                                   \
    "v_bfi_b32 v96, " MSK ", v96, v97\n"                                             \
    "v_add_u32 v97, " TK0 ", v96\n"                                                  \
    "v_bfi_b32 v98, " TK0 ", v97, -1\n"                                              \
    "v_lshl_or_b32 v98, v98, 1, 1\n"                                                 \
    "v_bitop3_b32 v98, v98, " TK0 ", v97 bitop3:0x60\n"                              \
    "v_ffbh_u32 " TK1 ", v98\n"                                                      \
    "v_lshlrev_b32 %0, " TK1 ", " TK0 "\n"                                           \
    "v_add_u32 v96, 1, v96\n"                                                        \
    "v_lshl_add_u32 %2, v96, " TK1 ", -1\n"

// ---- V2: x1 raw lives in v110 (pair v[110:111]), d %2; ops (q, b01, z, z); 11 VALU
#define V2_STEP(Q, B01, ZP, TKP, TK0, TK1)                                           \
    "v_mad_u64_u32 v[96:97], vcc, %2, " Q ", " ZP "\n"                               \
    "v_sub_u32 v98, %2, v97\n"                                                       \
    "v_mad_u64_u32 " TKP ", vcc, v98, " B01 ", v[110:111]\n"                         \
    "v_add_u32 v98, " TK0 ", v97\n"                                                  \
    "v_bfi_b32 v99, " TK0 ", v98, -1\n"                                              \
    "v_lshl_or_b32 v99, v99, 1, 1\n"                                                 \
    "v_bitop3_b32 v99, v99, " TK0 ", v98 bitop3:0x60\n"                              \
    "v_ffbh_u32 " TK1 ", v99\n"                                                      \
    "v_lshlrev_b32 v110, " TK1 ", " TK0 "\n"                                         \
    "v_add_u32 v97, 1, v97\n"                                                        \
    "v_lshl_add_u32 %2, v97, " TK1 ", -1\n"

// V0/V1 byte: ops of the byte in v[B..B+15] (4 x ds_read2st64_b64), tokens via 4 x ds_write2st64_b64
// %3 = ops LDS byte address of the chunk (this lane), %4 = token LDS byte address of the chunk
#define RD8_2(B, K)  /* read the 8 ops of byte K of the chunk into v[B..B+15]; step stride 512 B = st64 unit for b64 */ \
    "ds_read2st64_b64 v[" S(B) "+0:" S(B) "+3], %3 offset0:" S(K) "*8+0 offset1:" S(K) "*8+1\n"                  \
    "ds_read2st64_b64 v[" S(B) "+4:" S(B) "+7], %3 offset0:" S(K) "*8+2 offset1:" S(K) "*8+3\n"                  \
    "ds_read2st64_b64 v[" S(B) "+8:" S(B) "+11], %3 offset0:" S(K) "*8+4 offset1:" S(K) "*8+5\n"                 \
    "ds_read2st64_b64 v[" S(B) "+12:" S(B) "+15], %3 offset0:" S(K) "*8+6 offset1:" S(K) "*8+7\n"
#define WR2(K, J) "ds_write2st64_b64 %4, v[100:101], v[104:105] offset0:" S(K) "*8+" S(J) " offset1:" S(K) "*8+" S(J) "+1\n"
#define BYTE_2(STEP, B, K)                                                            \
    STEP("v[" S(B) "+0]", "v[" S(B) "+1]", "v100", "v101") STEP("v[" S(B) "+2]", "v[" S(B) "+3]", "v104", "v105") WR2(K, 0)   \
    STEP("v[" S(B) "+4]", "v[" S(B) "+5]", "v100", "v101") STEP("v[" S(B) "+6]", "v[" S(B) "+7]", "v104", "v105") WR2(K, 2)   \
    STEP("v[" S(B) "+8]", "v[" S(B) "+9]", "v100", "v101") STEP("v[" S(B) "+10]", "v[" S(B) "+11]", "v104", "v105") WR2(K, 4) \
    STEP("v[" S(B) "+12]", "v[" S(B) "+13]", "v100", "v101") STEP("v[" S(B) "+14]", "v[" S(B) "+15]", "v104", "v105") WR2(K, 6)

// V2 byte: ops 16 B/step -> 8 x ds_read_b128 into v[B..B+31]; op LDS layout [byte][step][lane] x 16 B: step stride 1024 B
#define RD8_4(B, K)                                                                                         \
    "ds_read_b128 v[" S(B) "+0:" S(B) "+3], %3 offset:" S(K) "*8192+0\n"                                     \
    "ds_read_b128 v[" S(B) "+4:" S(B) "+7], %3 offset:" S(K) "*8192+1024\n"                                  \
    "ds_read_b128 v[" S(B) "+8:" S(B) "+11], %3 offset:" S(K) "*8192+2048\n"                                 \
    "ds_read_b128 v[" S(B) "+12:" S(B) "+15], %3 offset:" S(K) "*8192+3072\n"                                \
    "ds_read_b128 v[" S(B) "+16:" S(B) "+19], %3 offset:" S(K) "*8192+4096\n"                                \
    "ds_read_b128 v[" S(B) "+20:" S(B) "+23], %3 offset:" S(K) "*8192+5120\n"                                \
    "ds_read_b128 v[" S(B) "+24:" S(B) "+27], %3 offset:" S(K) "*8192+6144\n"                                \
    "ds_read_b128 v[" S(B) "+28:" S(B) "+31], %3 offset:" S(K) "*8192+7168\n"
#define V2S(B, E, TKP, TK0, TK1) V2_STEP("v[" S(B) "+4*" S(E) "]", "v[" S(B) "+4*" S(E) "+1]", "v[" S(B) "+4*" S(E) "+2:" S(B) "+4*" S(E) "+3]", TKP, TK0, TK1)
#define BYTE_4(B, K)                                                                  \
    V2S(B, 0, "v[100:101]", "v100", "v101") V2S(B, 1, "v[104:105]", "v104", "v105") WR2(K, 0)  \
    V2S(B, 2, "v[100:101]", "v100", "v101") V2S(B, 3, "v[104:105]", "v104", "v105") WR2(K, 2)  \
    V2S(B, 4, "v[100:101]", "v100", "v101") V2S(B, 5, "v[104:105]", "v104", "v105") WR2(K, 4)  \
    V2S(B, 6, "v[100:101]", "v100", "v101") V2S(B, 7, "v[104:105]", "v104", "v105") WR2(K, 6)

// ---- V3 / V4: the shipped step (w3_coder4.h W3_X4_STEP): dn = hi32(d*q + (z:z)); x1n = x1 + (dn - d)*z
#define V3_STEP(Q, Z, ZP, TKP, TK0, TK1)                                             \
    "v_mad_u64_u32 v[96:97], vcc, %2, " Q ", " ZP "\n"                               \
    "v_sub_u32 v98, v97, %2\n"                                                       \
    "v_mad_u64_u32 " TKP ", vcc, v98, " Z ", v[110:111]\n"                           \
    "v_add_u32 v98, " TK0 ", v97\n"                                                  \
    "v_bfi_b32 v99, " TK0 ", v98, -1\n"                                              \
    "v_lshl_or_b32 v99, v99, 1, 1\n"                                                 \
    "v_bitop3_b32 v99, v99, " TK0 ", v98 bitop3:0x60\n"                              \
    "v_ffbh_u32 " TK1 ", v99\n"                                                      \
    "v_lshlrev_b32 v110, " TK1 ", " TK0 "\n"                                         \
    "v_add_u32 v97, 1, v97\n"                                                        \
    "v_lshl_add_u32 %2, v97, " TK1 ", -1\n"
#define RD8_3(B, K)                                                                                         \
    "ds_read_b96 v[" S(B) "+0:" S(B) "+2], %3 offset:" S(K) "*8192+0\n"                                      \
    "ds_read_b96 v[" S(B) "+4:" S(B) "+6], %3 offset:" S(K) "*8192+1024\n"                                   \
    "ds_read_b96 v[" S(B) "+8:" S(B) "+10], %3 offset:" S(K) "*8192+2048\n"                                  \
    "ds_read_b96 v[" S(B) "+12:" S(B) "+14], %3 offset:" S(K) "*8192+3072\n"                                 \
    "ds_read_b96 v[" S(B) "+16:" S(B) "+18], %3 offset:" S(K) "*8192+4096\n"                                 \
    "ds_read_b96 v[" S(B) "+20:" S(B) "+22], %3 offset:" S(K) "*8192+5120\n"                                 \
    "ds_read_b96 v[" S(B) "+24:" S(B) "+26], %3 offset:" S(K) "*8192+6144\n"                                 \
    "ds_read_b96 v[" S(B) "+28:" S(B) "+30], %3 offset:" S(K) "*8192+7168\n"
#define V3S(B, E, TKP, TK0, TK1) V3_STEP("v[" S(B) "+4*" S(E) "+2]", "v[" S(B) "+4*" S(E) "]", "v[" S(B) "+4*" S(E) ":" S(B) "+4*" S(E) "+1]", TKP, TK0, TK1)
#define BYTE_3(B, K)                                                                  \
    V3S(B, 0, "v[100:101]", "v100", "v101") V3S(B, 1, "v[104:105]", "v104", "v105") WR2(K, 0)  \
    V3S(B, 2, "v[100:101]", "v100", "v101") V3S(B, 3, "v[104:105]", "v104", "v105") WR2(K, 2)  \
    V3S(B, 4, "v[100:101]", "v100", "v101") V3S(B, 5, "v[104:105]", "v104", "v105") WR2(K, 4)  \
    V3S(B, 6, "v[100:101]", "v100", "v101") V3S(B, 7, "v[104:105]", "v104", "v105") WR2(K, 6)
// V4: group G (steps 4G .. 4G+3) of byte K in v[B+12G .. B+12G+11] = (z z z z | z z z z | q q q q) by three b128 reads (planes of 1 KiB)
#define RD8_5(B, K)                                                                                         \
    "ds_read_b128 v[" S(B) "+0:" S(B) "+3], %3 offset:" S(K) "*8192+0\n"                                     \
    "ds_read_b128 v[" S(B) "+4:" S(B) "+7], %3 offset:" S(K) "*8192+1024\n"                                  \
    "ds_read_b128 v[" S(B) "+8:" S(B) "+11], %3 offset:" S(K) "*8192+2048\n"                                 \
    "ds_read_b128 v[" S(B) "+12:" S(B) "+15], %3 offset:" S(K) "*8192+3072\n"                                \
    "ds_read_b128 v[" S(B) "+16:" S(B) "+19], %3 offset:" S(K) "*8192+4096\n"                                \
    "ds_read_b128 v[" S(B) "+20:" S(B) "+23], %3 offset:" S(K) "*8192+5120\n"
#define V4S(B, G, E, TKP, TK0, TK1) V3_STEP("v[" S(B) "+12*" S(G) "+8+" S(E) "]", "v[" S(B) "+12*" S(G) "+2*" S(E) "]", \
                                            "v[" S(B) "+12*" S(G) "+2*" S(E) ":" S(B) "+12*" S(G) "+2*" S(E) "+1]", TKP, TK0, TK1)
#define BYTE_5(B, K)                                                                  \
    V4S(B, 0, 0, "v[100:101]", "v100", "v101") V4S(B, 0, 1, "v[104:105]", "v104", "v105") WR2(K, 0)  \
    V4S(B, 0, 2, "v[100:101]", "v100", "v101") V4S(B, 0, 3, "v[104:105]", "v104", "v105") WR2(K, 2)  \
    V4S(B, 1, 0, "v[100:101]", "v100", "v101") V4S(B, 1, 1, "v[104:105]", "v104", "v105") WR2(K, 4)  \
    V4S(B, 1, 2, "v[100:101]", "v100", "v101") V4S(B, 1, 3, "v[104:105]", "v104", "v105") WR2(K, 6)

#define CLOB16 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
               "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95"
#define CLOBT "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v110","v111","vcc","memory"
#define CLOB64 "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
               "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"

// chunk of 4 bytes, ops double-buffered in v[64:79] / v[80:95] (V0, V1)
#define CHUNK4_2(STEP)                                                                 \
    RD8_2(64, 0) RD8_2(80, 1) "s_waitcnt lgkmcnt(4)\n" BYTE_2(STEP, 64, 0)              \
    RD8_2(64, 2) "s_waitcnt lgkmcnt(4)\n" BYTE_2(STEP, 80, 1)                           \
    RD8_2(80, 3) "s_waitcnt lgkmcnt(4)\n" BYTE_2(STEP, 64, 2)                           \
    "s_waitcnt lgkmcnt(0)\n" BYTE_2(STEP, 80, 3)
// the waits above: ops return in order; lgkmcnt(4) right after issuing 4 reads = everything older has landed

#define CHUNK4_4                                                                       \
    "v_mov_b32 v110, %0\n"                                                             \
    RD8_4(32, 0) RD8_4(64, 1) "s_waitcnt lgkmcnt(8)\n" BYTE_4(32, 0)                    \
    RD8_4(32, 2) "s_waitcnt lgkmcnt(8)\n" BYTE_4(64, 1)                                 \
    RD8_4(64, 3) "s_waitcnt lgkmcnt(8)\n" BYTE_4(32, 2)                                 \
    "s_waitcnt lgkmcnt(0)\n" BYTE_4(64, 3)                                              \
    "v_mov_b32 %0, v110\n"

#define CHUNK4_3                                                                       \
    "v_mov_b32 v110, %0\n"                                                             \
    RD8_3(32, 0) RD8_3(64, 1) "s_waitcnt lgkmcnt(8)\n" BYTE_3(32, 0)                    \
    RD8_3(32, 2) "s_waitcnt lgkmcnt(8)\n" BYTE_3(64, 1)                                 \
    RD8_3(64, 3) "s_waitcnt lgkmcnt(8)\n" BYTE_3(32, 2)                                 \
    "s_waitcnt lgkmcnt(0)\n" BYTE_3(64, 3)                                              \
    "v_mov_b32 %0, v110\n"
#define CHUNK4_5                                                                       \
    "v_mov_b32 v110, %0\n"                                                             \
    RD8_5(32, 0) RD8_5(64, 1) "s_waitcnt lgkmcnt(6)\n" BYTE_5(32, 0)                    \
    RD8_5(32, 2) "s_waitcnt lgkmcnt(6)\n" BYTE_5(64, 1)                                 \
    RD8_5(64, 3) "s_waitcnt lgkmcnt(6)\n" BYTE_5(32, 2)                                 \
    "s_waitcnt lgkmcnt(0)\n" BYTE_5(64, 3)                                              \
    "v_mov_b32 %0, v110\n"

#define RING 8u   // ring depth in input bytes

struct Res { uint32_t x1, x2, d, cs; };

// reference step in plain C (k_coder_fast's formulas) for the checksum
__device__ __forceinline__ void ref_step(uint32_t &x1, uint32_t &x2, uint32_t p32, uint32_t mask, uint32_t &tx, uint32_t &ts) {
    const uint32_t xmid = x1 + __umulhi(x2 - x1, p32);
    x1 = (x1 & mask) | ((xmid + 1u) & ~mask);
    x2 = (xmid & mask) | (x2 & ~mask);
    const uint32_t s = (uint32_t)__builtin_clz((x1 ^ x2) & (((~x1 | x2) << 1) | 1u));
    tx = x1; ts = s;
    x1 = (x1 << s) & 0x7FFFFFFFu;
    x2 = (((x2 + 1u) << s) - 1u) | 0x80000000u;
}

// ops are generated by this hash so host/device agree without a buffer
__device__ __forceinline__ uint32_t mixh(uint32_t a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }
__device__ __forceinline__ void gen_op(uint32_t lane, uint32_t ringbyte, uint32_t j, uint32_t &p32, uint32_t &mask) {
    const uint32_t h = mixh(lane * 977u + ringbyte * 131u + j * 7u + 12345u);
    uint32_t p = h & 0xFFFFu; if (p == 0u) p = 1u;
    if ((h >> 16) & 1u) p = (p >> 5) | 1u;            // a share of very skewed probabilities (long E3 / shift cases)
    p32 = p << 16;
    mask = ((h >> 20) & 0xFFu) * 65536u / 256u < p ? 0xFFFFFFFFu : 0u;   // bit drawn roughly according to p
}

template <int V, bool CHECK>
__global__ void __launch_bounds__(64) k_xstep(Res *res, Stamp *st, uint32_t nbytes, int prio) {
    extern __shared__ uint8_t lds_raw[];
    // V0/V1: ops [RING][8][64] x 8 B = 32 KiB, tokens [RING][8][64] x 8 B = 32 KiB;  V2: ops x 16 B = 64 KiB
    constexpr uint32_t OPB = V >= 2 ? 16u : 8u;
    uint8_t *opq = lds_raw;
    uint8_t *tok = lds_raw + RING * 8u * 64u * OPB;
    const uint32_t lane = threadIdx.x;
    for (uint32_t rb = 0; rb < RING; rb++)
        for (uint32_t j = 0; j < 8; j++) {
            uint32_t p32, mask; gen_op(lane + blockIdx.x * 64u, rb, j, p32, mask);
            uint8_t *o = opq + ((rb * 8u + j) * 64u + lane) * OPB;
            if (V == 3 || V == 4) {
                const uint32_t z = ~mask, q = (p32 ^ z) + (z & 1u);   // bit ? p32 : 2^32 - p32
                uint32_t *ob = (uint32_t *)(opq + (rb * 8u * 64u) * OPB);   // this ring byte's 8 KiB: [step or plane][lane] x 16 B
                if (V == 3) { uint32_t *o3 = ob + (j * 64u + lane) * 4u; o3[0] = z; o3[1] = z; o3[2] = q; o3[3] = 0u; }
                else {
                    const uint32_t g = j >> 2, e = j & 3u;
                    ob[((3u * g + (e >> 1)) * 64u + lane) * 4u + 2u * (e & 1u)] = z;
                    ob[((3u * g + (e >> 1)) * 64u + lane) * 4u + 2u * (e & 1u) + 1u] = z;
                    ob[((3u * g + 2u) * 64u + lane) * 4u + e] = q;
                }
            } else if (V == 2) {
                const uint32_t z = ~mask;
                ((uint32_t *)o)[0] = (p32 ^ z) + (z & 1u);   // bit ? p32 : 2^32 - p32
                ((uint32_t *)o)[1] = z & 1u;
                ((uint32_t *)o)[2] = z; ((uint32_t *)o)[3] = z;
            } else { ((uint32_t *)o)[0] = p32; ((uint32_t *)o)[1] = mask; }
        }
    __syncthreads();
    if (prio) __builtin_amdgcn_s_setprio(3);
    uint32_t x1 = 0u, x2 = 0xFFFFFFFFu, d = 0xFFFFFFFFu, cs = 0u;
    uint32_t rx1 = 0u, rx2 = 0xFFFFFFFFu, rcs = 0u;
    const uint32_t op_lane = (uint32_t)(uintptr_t)(opq + lane * OPB) - (uint32_t)(uintptr_t)lds_raw;
    const uint32_t tk_lane = (uint32_t)(uintptr_t)(tok + lane * 8u) - (uint32_t)(uintptr_t)lds_raw;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds_raw;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (uint32_t i = 0; i < nbytes; i += 4) {
        const uint32_t rb = i & (RING - 1u);
        uint32_t opa = lds_base + op_lane + rb * 8u * 64u * OPB;
        uint32_t tka = lds_base + tk_lane + rb * 8u * 64u * 8u;
        if (V == 0) asm volatile(CHUNK4_2(V0_STEP) : "+v"(x1), "+v"(x2), "+v"(d) : "v"(opa), "v"(tka) : CLOB16, CLOBT);
        if (V == 1) asm volatile(CHUNK4_2(V1_STEP) : "+v"(x1), "+v"(x2), "+v"(d) : "v"(opa), "v"(tka) : CLOB16, CLOBT);
        if (V == 2) asm volatile(CHUNK4_4 : "+v"(x1), "+v"(x2), "+v"(d) : "v"(opa), "v"(tka) : CLOB64, CLOB16, CLOBT);
        if (V == 3) asm volatile(CHUNK4_3 : "+v"(x1), "+v"(x2), "+v"(d) : "v"(opa), "v"(tka) : CLOB64, CLOB16, CLOBT);
        if (V == 4) asm volatile(CHUNK4_5 : "+v"(x1), "+v"(x2), "+v"(d) : "v"(opa), "v"(tka) : CLOB64, CLOB16, CLOBT);
        if (CHECK) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (uint32_t k = 0; k < 4; k++)
                for (uint32_t j = 0; j < 8; j++) {
                    const uint2 t = *(const uint2 *)(tok + (((rb + k) * 8u + j) * 64u + lane) * 8u);
                    cs = cs * 31u + (t.x & 0x7FFFFFFFu) * 7u + t.y;
                    uint32_t p32, mask, tx, ts; gen_op(lane + blockIdx.x * 64u, rb + k, j, p32, mask);
                    ref_step(rx1, rx2, p32, mask, tx, ts);
                    rcs = rcs * 31u + (tx & 0x7FFFFFFFu) * 7u + ts;
                }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (CHECK) {
        Res r; r.x1 = (x1 & 0x7FFFFFFFu) ^ rx1; r.d = (V == 0 ? x2 - x1 : d) ^ (rx2 - rx1); r.x2 = 0; r.cs = cs ^ rcs;
        res[blockIdx.x * 64u + lane] = r;
    } else if (lane == 0) { Res r; r.x1 = x1; r.x2 = x2; r.d = d; r.cs = 0; res[blockIdx.x * 64u] = r; }
    if (lane == 0 && blockIdx.x == 0) { st->t0 = t0; st->t1 = t1; st->r0 = r0; st->r1 = r1; }
}

template <int V>
static void run_variant(const char *name, Res *d_res, Stamp *d_st, int wgs, uint32_t nbytes, int prio) {
    const size_t lds = (size_t)RING * 8 * 64 * (V >= 2 ? 16 : 8) + (size_t)RING * 8 * 64 * 8;
    CK(hipFuncSetAttribute((const void *)k_xstep<V, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_xstep<V, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // correctness against the C reference step
    hipLaunchKernelGGL((k_xstep<V, true>), dim3(2), dim3(64), lds, 0, d_res, d_st, 4096u, 0);
    CK(hipDeviceSynchronize());
    std::vector<Res> h(128);
    CK(hipMemcpy(h.data(), d_res, 128 * sizeof(Res), hipMemcpyDeviceToHost));
    int bad = 0;
    for (auto &r : h) if (r.x1 | r.d | r.cs) bad++;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_xstep<V, false>), dim3(wgs), dim3(64), lds, 0, d_res, d_st, nbytes / 8, prio);   // warm-up
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_xstep<V, false>), dim3(wgs), dim3(64), lds, 0, d_res, d_st, nbytes, prio);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Stamp s; CK(hipMemcpy(&s, d_st, sizeof(s), hipMemcpyDeviceToHost));
    const double steps = (double)nbytes * 8.0;
    printf("%-34s wgs=%3d prio=%d lds=%6zu  mismatching lanes %3d/128 | %7.2f ticks/step  %6.2f ns/step (in-kernel)  %6.2f ns/step (events)  clock %.0f MHz\n",
           name, wgs, prio, lds, bad, (double)(s.t1 - s.t0) / steps, (double)(s.r1 - s.r0) * 10.0 / steps, ms * 1e6 / steps,
           (double)(s.t1 - s.t0) / ((double)(s.r1 - s.r0) * 10.0) * 1e3);
}

int main() {
    uint32_t *out; Stamp *st; Res *res;
    CK(hipMalloc(&out, 1024)); CK(hipMalloc(&st, sizeof(Stamp))); CK(hipMalloc(&res, 1024 * 64 * sizeof(Res)));
    struct { const char *n; void (*k)(uint32_t *, Stamp *, uint32_t, uint32_t); int per_iter; } T[] = {
        {"v_add_u32 dependent", p_dep1, 1}, {"2 independent v_add chains", p_ind2, 2}, {"4 independent v_add chains", p_ind4, 4},
        {"8 independent v_add chains", p_ind8, 8}, {"v_add3_u32 (VOP3) dependent", p_dep1_vop3, 1}, {"4 independent v_add3 (VOP3)", p_ind4_vop3, 4},
        {"v_add_u32 dependent, s_setprio 3", p_dep1_prio, 1}, {"v_mul_hi_u32 dependent", p_dep_mulhi, 1},
        {"v_mad_u64_u32 dependent", p_dep_mad64, 1}, {"v_add + ds_write_b32", p_dep_add_dswrite, 2}, {"4 dep v_add + ds_write_b32", p_dep4_dswrite, 5},
        {"4 dep v_add + ds_read_b32+wait", p_dep4_dsread, 5}, {"4 dep v_add + s_nop", p_dep4_snop, 5}, {"4 dep v_add + s_add_u32", p_dep4_salu, 5}};
    for (auto &t : T) {
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, out, st, 3u, 5u);
        CK(hipDeviceSynchronize());
        Stamp s; CK(hipMemcpy(&s, st, sizeof(s), hipMemcpyDeviceToHost));
        const double n = (double)REP * t.per_iter;
        printf("%-36s %6.2f ticks/instr  %6.2f ns/instr  (%d instr per iteration; clock %.0f MHz)\n", t.n, (double)(s.t1 - s.t0) / n,
               (double)(s.r1 - s.r0) * 10.0 / n, t.per_iter, (double)(s.t1 - s.t0) / ((double)(s.r1 - s.r0) * 10.0) * 1e3);
    }
    const uint32_t nbytes = 65536;
    for (int wgs : {1, 239}) {
        for (int prio : {0, 1}) {
            run_variant<0>("V0 shipped 15 VALU (x1,x2,d)", res, st, wgs, nbytes, prio);
            run_variant<1>("V1 13 VALU (x1 raw, d)", res, st, wgs, nbytes, prio);
            run_variant<2>("V2 11 VALU, 2 x mad_u64_u32", res, st, wgs, nbytes, prio);
            run_variant<3>("V3 shipped x4 step, b96 per step", res, st, wgs, nbytes, prio);
            run_variant<4>("V4 x4 step, 3 x b128 per 4 steps", res, st, wgs, nbytes, prio);
        }
    }
    return 0;
}
