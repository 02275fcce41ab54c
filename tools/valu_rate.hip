// valu_rate.hip - what is the vector-ALU issue peak of gfx950 for the instructions the front end is made of?
//
// Each wave runs ITERS x 32 INDEPENDENT copies of one instruction (8 accumulator chains x 4), brackets them with s_memtime and
// records its hardware SIMD (HW_ID / XCC_ID).  The grid is sized so that W = 1, 2, 4, 8 waves share a SIMD; the host keeps the
// SIMDs that really held W waves and prints wave-instructions per SIMD-cycle = W x N / mean(elapsed).  1/rate is the issue cost of
// one wave64 instruction in cycles: 2 means the SIMD processes 32 lanes per cycle, 4 means 16.
// Build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > gpurun_out/valu_rate.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITERS = 4096;          // x 32 instructions = 131 072 per wave

struct rec { uint64_t cyc; uint32_t hw, xcc; };

#define BODY8(INS) \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

// One kernel per instruction: OPSTR is the asm template with %0 = accumulator (read-modify), %1, %2 = loop-invariant operands.
#define DEF_KERNEL(NAME, OPSTR)                                                                                              \
__global__ void k_##NAME(rec* out, uint32_t seed)                                                                            \
{                                                                                                                            \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
    uint32_t b = seed * 0x9e3779b9u + threadIdx.x, c = seed ^ 0x01020304u;                                                  \
    __syncthreads();                                                                                                         \
    uint64_t t0, t1;                                                                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");                                             \
    for (int i = 0; i < ITERS; i++) {                                                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                                      \
            asm volatile(OPSTR(0) "\n\t" OPSTR(1) "\n\t" OPSTR(2) "\n\t" OPSTR(3) "\n\t" OPSTR(4) "\n\t" OPSTR(5) "\n\t" OPSTR(6) "\n\t" OPSTR(7) \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc"); \
        }                                                                                                                    \
    }                                                                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                                             \
    uint32_t hw, xcc;                                                                                                        \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                                         \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                                       \
    if ((threadIdx.x & 63) == 0) {                                                                                           \
        rec r; r.cyc = t1 - t0; r.hw = hw; r.xcc = xcc;                                                                      \
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;                                                               \
    }                                                                                                                        \
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0].hw = 1;                                                  \
}

#define S_(x) #x
#define OP_ADD(i)      "v_add_u32 %" S_(i) ", %" S_(i) ", %8"
#define OP_AND(i)      "v_and_b32 %" S_(i) ", %" S_(i) ", %8"
#define OP_FMA(i)      "v_fma_f32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_PKFMA16(i)  "v_pk_fma_f16 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_PKADD(i)    "v_pk_add_i16 %" S_(i) ", %" S_(i) ", %8"
#define OP_PKSUB(i)    "v_pk_sub_i16 %" S_(i) ", %" S_(i) ", %8"
#define OP_PKMIN(i)    "v_pk_min_i16 %" S_(i) ", %" S_(i) ", %8"
#define OP_PKMAX(i)    "v_pk_max_i16 %" S_(i) ", %" S_(i) ", %8"
#define OP_PKMIN3(i)   "v_pk_minimum3_f16 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_PKMAX3(i)   "v_pk_maximum3_f16 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_PERM(i)     "v_perm_b32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_LSHLOR(i)   "v_lshl_or_b32 %" S_(i) ", %" S_(i) ", 3, %8"
#define OP_MAD24(i)    "v_mad_i32_i24 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_MADU16(i)   "v_pk_mad_u16 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_PKMUL(i)    "v_pk_mul_lo_u16 %" S_(i) ", %" S_(i) ", %8"
#define OP_DOT4(i)     "v_dot4_u32_u8 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_DOT2(i)     "v_dot2_u32_u16 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_BFE(i)      "v_bfe_u32 %" S_(i) ", %" S_(i) ", 3, 9"
#define OP_MULLO(i)    "v_mul_lo_u32 %" S_(i) ", %" S_(i) ", %8"
#define OP_MUL24(i)    "v_mul_u32_u24 %" S_(i) ", %" S_(i) ", %8"
#define OP_DPP(i)      "v_mov_b32_dpp %" S_(i) ", %" S_(i) " row_shr:1 row_mask:0xf bank_mask:0xf"
#define OP_ADDDPP(i)   "v_add_u32_dpp %" S_(i) ", %" S_(i) ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"
#define OP_SDWA(i)     "v_add_u32_sdwa %" S_(i) ", %" S_(i) ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
#define OP_MAX3(i)     "v_max3_i32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_MED3(i)     "v_med3_i32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_ALIGNBIT(i) "v_alignbit_b32 %" S_(i) ", %" S_(i) ", %8, 8"
#define OP_BCNT(i)     "v_bcnt_u32_b32 %" S_(i) ", %" S_(i) ", %8"
#define OP_CMP(i)      "v_cmp_lt_u32 vcc, %" S_(i) ", %8"
#define OP_SAD(i)      "v_sad_u8 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_MSAD(i)     "v_msad_u8 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_PKLSHR(i)   "v_pk_lshrrev_b16 %" S_(i) ", 3, %" S_(i)
#define OP_PKASHR(i)   "v_pk_ashrrev_i16 %" S_(i) ", 3, %" S_(i)
#define OP_XAD(i)      "v_xad_u32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_AND_OR(i)   "v_and_or_b32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_ADD3(i)     "v_add3_u32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_LSHL(i)   "v_lshlrev_b32 %" S_(i) ", 1, %" S_(i)
#define OP_CVTPK(i)    "v_cvt_pk_u8_f32 %" S_(i) ", %" S_(i) ", 1, %8"

#define OP_OR(i)       "v_or_b32 %" S_(i) ", %" S_(i) ", %8"
#define OP_XOR(i)      "v_xor_b32 %" S_(i) ", %" S_(i) ", %8"
#define OP_SUB(i)      "v_sub_u32 %" S_(i) ", %" S_(i) ", %8"
#define OP_MINU(i)     "v_min_u32 %" S_(i) ", %" S_(i) ", %8"
#define OP_MAXI(i)     "v_max_i32 %" S_(i) ", %" S_(i) ", %8"
#define OP_LSHR(i)     "v_lshrrev_b32 %" S_(i) ", 1, %" S_(i)
#define OP_ASHR(i)     "v_ashrrev_i32 %" S_(i) ", 1, %" S_(i)
#define OP_CNDMASK(i)  "v_cndmask_b32 %" S_(i) ", %" S_(i) ", %8, vcc"
#define OP_MOV(i)      "v_mov_b32 %" S_(i) ", %8"
#define OP_ADDF(i)     "v_add_f32 %" S_(i) ", %" S_(i) ", %8"
#define OP_MULF(i)     "v_mul_f32 %" S_(i) ", %" S_(i) ", %8"
#define OP_MAXF(i)     "v_max_f32 %" S_(i) ", %" S_(i) ", %8"
#define OP_ADDCO(i)    "v_add_co_u32 %" S_(i) ", vcc, %" S_(i) ", %8"
#define OP_ADDU16(i)   "v_add_u16 %" S_(i) ", %" S_(i) ", %8"
#define OP_MINU16(i)   "v_min_u16 %" S_(i) ", %" S_(i) ", %8"
#define OP_MAXI16(i)   "v_max_i16 %" S_(i) ", %" S_(i) ", %8"
#define OP_ADDF16(i)   "v_add_f16 %" S_(i) ", %" S_(i) ", %8"
#define OP_MADU24(i)   "v_mad_u32_u24 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_LSHLADD(i)  "v_lshl_add_u32 %" S_(i) ", %" S_(i) ", 2, %8"
#define OP_ADDLSHL(i)  "v_add_lshl_u32 %" S_(i) ", %" S_(i) ", %8, 2"
#define OP_OR3(i)      "v_or3_b32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_BFI(i)      "v_bfi_b32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_PKADDU16(i) "v_pk_add_u16 %" S_(i) ", %" S_(i) ", %8"
#define OP_PKADDF16(i) "v_pk_add_f16 %" S_(i) ", %" S_(i) ", %8"
#define OP_CVTUB0(i)   "v_cvt_f32_ubyte0 %" S_(i) ", %" S_(i)
#define OP_MIN3I(i)    "v_min3_i32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_MAXF3(i)    "v_max3_f32 %" S_(i) ", %" S_(i) ", %8, %9"
#define OP_SUBREV(i)   "v_subrev_u32 %" S_(i) ", %" S_(i) ", %8"
#define OP_FMAC(i)     "v_fmac_f32 %" S_(i) ", %8, %9"

// alternating fast-class / packed instructions on independent chains: one pipe (costs add) or two (the larger one)?
#define OP_MIX(i) OP_MIX_##i
#define OP_MIX_0 "v_add_u32 %0, %0, %8"
#define OP_MIX_1 "v_pk_add_i16 %1, %1, %8"
#define OP_MIX_2 "v_add_u32 %2, %2, %8"
#define OP_MIX_3 "v_pk_add_i16 %3, %3, %8"
#define OP_MIX_4 "v_add_u32 %4, %4, %8"
#define OP_MIX_5 "v_pk_add_i16 %5, %5, %8"
#define OP_MIX_6 "v_add_u32 %6, %6, %8"
#define OP_MIX_7 "v_pk_add_i16 %7, %7, %8"

#define LIST(X) X(ADD) X(AND) X(FMA) X(PKFMA16) X(PKADD) X(PKSUB) X(PKMIN) X(PKMAX) X(PKMIN3) X(PKMAX3) X(PERM) X(LSHLOR) X(MAD24) X(MADU16) \
    X(PKMUL) X(DOT4) X(DOT2) X(BFE) X(MULLO) X(MUL24) X(DPP) X(ADDDPP) X(SDWA) X(MAX3) X(MED3) X(ALIGNBIT) X(BCNT) X(CMP) X(SAD) X(MSAD) \
    X(PKLSHR) X(PKASHR) X(XAD) X(AND_OR) X(ADD3) X(LSHL) X(CVTPK) \
    X(OR) X(XOR) X(SUB) X(MINU) X(MAXI) X(LSHR) X(ASHR) X(CNDMASK) X(MOV) X(ADDF) X(MULF) X(MAXF) X(ADDCO) X(ADDU16) X(MINU16) X(MAXI16) X(ADDF16) \
    X(MADU24) X(LSHLADD) X(ADDLSHL) X(OR3) X(BFI) X(PKADDU16) X(PKADDF16) X(CVTUB0) X(MIN3I) X(MAXF3) X(SUBREV) X(FMAC) X(MIX)

#define X(N) DEF_KERNEL(N, OP_##N)
LIST(X)
#undef X

typedef void (*kfn)(rec*, uint32_t);
struct ent { const char* name; kfn fn; };

int main(int argc, char** argv)
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    std::vector<ent> ks = {
#define X(N) { #N, k_##N },
        LIST(X)
#undef X
    };
    rec* d_out; const int max_waves = ncu * 4 * 8 * 2;
    CK(hipMalloc(&d_out, sizeof(rec) * max_waves));
    std::vector<rec> h(max_waves);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double ninstr = (double)ITERS * 32;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"instr_per_wave\": %.0f, \"results\": {\n", prop.gcnArchName, ncu, prop.clockRate, ninstr);
    for (size_t ki = 0; ki < ks.size(); ki++) {
        printf(" \"%s\": {", ks[ki].name);
        const int Ws[4] = { 1, 2, 4, 8 };
        for (int wi = 0; wi < 4; wi++) {
            const int W = Ws[wi];
            // W <= 4: one workgroup of 256 W threads per CU (4 W waves -> W per SIMD); W = 8: two workgroups of 1024 per CU
            const int bs = W <= 4 ? 256 * W : 1024, grid = W <= 4 ? ncu : 2 * ncu, waves = grid * bs / 64;
            CK(hipMemset(d_out, 0, sizeof(rec) * max_waves));
            hipLaunchKernelGGL(ks[ki].fn, dim3(grid), dim3(bs), 0, 0, d_out, 12345u + wi);      // warm-up (code load, clocks)
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(ks[ki].fn, dim3(grid), dim3(bs), 0, 0, d_out, 777u + wi);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d_out, sizeof(rec) * waves, hipMemcpyDeviceToHost));
            // group the waves by the SIMD they ran on: se / sh / cu / simd of HW_ID (bits 4-5 simd, 8-11 cu, 12 sh, 13-15 se) + XCC
            std::map<uint64_t, std::vector<uint64_t>> by;
            for (int w = 0; w < waves; w++) by[((uint64_t)(h[w].xcc & 0xf) << 32) | (h[w].hw & 0xff30u)].push_back(h[w].cyc);
            double sum = 0; long cnt = 0; int simds_exact = 0;
            for (auto& kv : by) if ((int)kv.second.size() == W) { simds_exact++; for (uint64_t c : kv.second) { sum += (double)c; cnt++; } }
            const double mean_cyc = cnt ? sum / cnt : 0;
            const double rate = mean_cyc > 0 ? W * ninstr / mean_cyc : 0;          // wave-instructions per SIMD-cycle (s_memtime ticks)
            const double wall_rate = (double)waves * ninstr / (ms * 1e-3);         // chip-wide wave-instructions per second (includes launch)
            printf("%s\"w%d\": {\"per_simd_cycle\": %.4f, \"cycles_per_instr\": %.3f, \"simds_with_W\": %d, \"simds_used\": %zu, \"chip_winstr_per_s\": %.4g, \"ns_per_instr_per_simd\": %.4f, \"ms\": %.4f}",
                   wi ? ", " : "", W, rate, rate > 0 ? 1.0 / rate : 0, simds_exact, by.size(), wall_rate, ms * 1e6 / (W * ninstr), ms);
        }
        printf("}%s\n", ki + 1 < ks.size() ? "," : "");
        fflush(stdout);
    }
    printf("}}\n");
    return 0;
}
