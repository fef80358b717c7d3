// valu_issue.hip -- how many cycles does a wave64 VALU instruction hold a gfx950 SIMD for, and does v_pk_fma_f32 double the fp32 rate?
// (VERDICT r2 "next" #2c: MI355X_MICROARCH.md says 2 cycles (SIMD-32); the stepper's counters / env-count sweep said 4.)
// Standalone: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
// Every kernel = one wavefront per workgroup (like the stepper) running R repetitions of a 64-instruction unrolled block of ONE
// instruction kind, either as a single dependent chain or as 8 independent chains; s_memtime around the loop gives the wave's own
// cycles, HIP events around the launch the wall time.  Reported per (kind, wavefronts per SIMD):
//   wave cyc/instr = mean over wavefronts of (s_memtime delta) / instructions           -- what ONE wavefront sees
//   SIMD cyc/instr = wall * f_shader * 1024 SIMDs / (wavefronts * instructions)            -- the issue cost when the SIMDs are full
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum Kind { FMA_DEP, FMA_IND, PKFMA_DEP, PKFMA_IND, PKMUL_IND, PKADD_IND, MUL_IND, ADD_DPP_IND, FMAC_DPP_IND, MOV_DPP_IND, RCP_IND, SIN_IND, SQRT_IND, CNDMASK_IND,
            LDS_RT, BPERMUTE_DEP, LDS_RD128, BARRIER, FMA_2CH, FMA_4CH, PKFMA_2CH, CNDMASK_SGPR, FMA_SGPR, MUL_SGPR, CMP_VCC, CMP_SGPR, FMAAK, MOV_IND, FMA_CND_MIX, FMA_BANK3, FMA_BANK2, FMAC_IND, FMAC_BANK, SUB_IND, NKINDS };
static const char* kind_name[NKINDS] = {"v_fma_f32 dependent chain", "v_fma_f32 8 independent chains", "v_pk_fma_f32 dependent chain", "v_pk_fma_f32 8 independent chains",
    "v_pk_mul_f32 8 independent", "v_pk_add_f32 8 independent", "v_mul_f32 (VOP2) 8 independent", "v_add_f32 dpp row_shr:1 8 independent", "v_fmac_f32 dpp row_shr:1 8 independent",
    "v_mov_b32 dpp row_shr:1 8 independent", "v_rcp_f32 8 independent", "v_sin_f32 8 independent", "v_sqrt_f32 8 independent", "v_cndmask_b32 8 independent",
    "ds_write_b32 -> ds_read_b32 round trip (dependent)", "ds_bpermute_b32 dependent chain", "ds_read_b128 8 in flight", "s_barrier (1-wave workgroup)",
    "v_fma_f32 2 interleaved chains", "v_fma_f32 4 interleaved chains", "v_pk_fma_f32 2 interleaved chains",
    "v_cndmask_b32_e64 (SGPR-pair mask) 8 independent", "v_fma_f32 with one SGPR source, 8 independent", "v_mul_f32 with an SGPR source, 8 independent",
    "v_cmp_lt_f32_e32 (writes vcc)", "v_cmp_lt_f32_e64 (writes an SGPR pair)", "v_fmaak_f32 (literal), 8 independent", "v_mov_b32 8 independent", "v_fma_f32 / v_cndmask_b32 (vcc) alternating",
    "v_fma_f32, three sources in ONE VGPR bank (v8 v12 v16)", "v_fma_f32, two sources in one bank", "v_fmac_f32_e32 8 independent (distinct banks)",
    "v_fmac_f32_e32, both sources and dst in one bank", "v_sub_f32_e32 8 independent"};

template <int K>
__global__ __launch_bounds__(64) void k_bench(float* out, unsigned long long* cyc, int reps, float seed) {
    __shared__ float lds[64 * 8];
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 0.999f, c = 0.001f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const f2 pb = {b, b}, pc = {c, c};
    int addr = threadIdx.x * 4, perm = ((threadIdx.x + 1) & 63) * 4;
    f4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0, q6 = q0, q7 = q0;
    lds[threadIdx.x] = a0;
    for (int i = 0; i < 8; ++i) lds[64 * i + threadIdx.x] = a0 + i;
    __syncthreads();
    asm volatile("s_mov_b32 s20, 0x3f7fbe77\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "s20", "s21", "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (K == FMA_DEP) { asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c)); }
        if (K == FMA_2CH) { asm volatile(REP8(REP8("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")) : "+v"(a0), "+v"(a1) : "v"(b), "v"(c)); }
        if (K == FMA_4CH) { asm volatile(REP8(REP8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c)); }
#define IND8(OP) asm volatile(REP8(OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n") \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))
        if (K == FMA_IND) { IND8("v_fma_f32"); }
        if (K == PKFMA_DEP) { asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p0) : "v"(pb), "v"(pc)); }
        if (K == PKFMA_2CH) { asm volatile(REP8(REP8("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n")) : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc)); }
#define PIND8(OP, TAIL) asm volatile(REP8(OP " %0, %0, %8" TAIL "\n" OP " %1, %1, %8" TAIL "\n" OP " %2, %2, %8" TAIL "\n" OP " %3, %3, %8" TAIL "\n" OP " %4, %4, %8" TAIL "\n" OP " %5, %5, %8" TAIL "\n" OP " %6, %6, %8" TAIL "\n" OP " %7, %7, %8" TAIL "\n") \
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc))
        if (K == PKFMA_IND) { PIND8("v_pk_fma_f32", ", %9"); }
        if (K == PKMUL_IND) { PIND8("v_pk_mul_f32", ""); }
        if (K == PKADD_IND) { PIND8("v_pk_add_f32", ""); }
#define SIND8(OP, TAIL) asm volatile(REP8(OP " %0, %0, %8 " TAIL "\n" OP " %1, %1, %8 " TAIL "\n" OP " %2, %2, %8 " TAIL "\n" OP " %3, %3, %8 " TAIL "\n" OP " %4, %4, %8 " TAIL "\n" OP " %5, %5, %8 " TAIL "\n" OP " %6, %6, %8 " TAIL "\n" OP " %7, %7, %8 " TAIL "\n") \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
        if (K == MUL_IND) { SIND8("v_mul_f32_e32", ""); }
        if (K == ADD_DPP_IND) { SIND8("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"); }
        if (K == FMAC_DPP_IND) { SIND8("v_fmac_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"); }
#define UIND8(OP, TAIL) asm volatile(REP8(OP " %0, %0 " TAIL "\n" OP " %1, %1 " TAIL "\n" OP " %2, %2 " TAIL "\n" OP " %3, %3 " TAIL "\n" OP " %4, %4 " TAIL "\n" OP " %5, %5 " TAIL "\n" OP " %6, %6 " TAIL "\n" OP " %7, %7 " TAIL "\n") \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
        if (K == MOV_DPP_IND) { UIND8("v_mov_b32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"); }
        if (K == RCP_IND) { UIND8("v_rcp_f32_e32", ""); }
        if (K == SIN_IND) { UIND8("v_sin_f32_e32", ""); }
        if (K == SQRT_IND) { UIND8("v_sqrt_f32_e32", ""); }
        if (K == CNDMASK_IND) { SIND8("v_cndmask_b32_e32", ", vcc"); }
        if (K == LDS_RT) { asm volatile(REP8("ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a0) : "v"(addr) : "memory"); }
        if (K == BPERMUTE_DEP) { asm volatile(REP8("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a0) : "v"(perm) : "memory"); }
        if (K == LDS_RD128) {
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                         "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr * 4 % 1024) : "memory");
            a0 += q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x;
        }
        if (K == BARRIER) { asm volatile(REP8("s_barrier\n") ::: "memory"); }
        if (K == FMA_BANK3) { asm volatile(REP8("v_fma_f32 v20, v8, v12, v16\n v_fma_f32 v21, v8, v12, v16\n v_fma_f32 v22, v8, v12, v16\n v_fma_f32 v23, v8, v12, v16\n v_fma_f32 v24, v8, v12, v16\n v_fma_f32 v25, v8, v12, v16\n v_fma_f32 v26, v8, v12, v16\n v_fma_f32 v27, v8, v12, v16\n")
            ::: "v8", "v12", "v16", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"); }
        if (K == FMA_BANK2) { asm volatile(REP8("v_fma_f32 v20, v8, v12, v17\n v_fma_f32 v21, v8, v12, v17\n v_fma_f32 v22, v8, v12, v17\n v_fma_f32 v23, v8, v12, v17\n v_fma_f32 v24, v8, v12, v17\n v_fma_f32 v25, v8, v12, v17\n v_fma_f32 v26, v8, v12, v17\n v_fma_f32 v27, v8, v12, v17\n")
            ::: "v8", "v12", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"); }
        if (K == FMAC_IND) { asm volatile(REP8("v_fmac_f32_e32 v20, v9, v14\n v_fmac_f32_e32 v21, v10, v15\n v_fmac_f32_e32 v22, v11, v12\n v_fmac_f32_e32 v23, v8, v13\n v_fmac_f32_e32 v24, v9, v14\n v_fmac_f32_e32 v25, v10, v15\n v_fmac_f32_e32 v26, v11, v12\n v_fmac_f32_e32 v27, v8, v13\n")
            ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"); }
        if (K == FMAC_BANK) { asm volatile(REP8("v_fmac_f32_e32 v20, v8, v12\n v_fmac_f32_e32 v24, v8, v12\n v_fmac_f32_e32 v28, v8, v12\n v_fmac_f32_e32 v32, v8, v12\n v_fmac_f32_e32 v36, v8, v12\n v_fmac_f32_e32 v40, v8, v12\n v_fmac_f32_e32 v44, v8, v12\n v_fmac_f32_e32 v48, v8, v12\n")
            ::: "v8", "v12", "v20", "v24", "v28", "v32", "v36", "v40", "v44", "v48"); }
        if (K == SUB_IND) { SIND8("v_sub_f32_e32", ""); }
        if (K == CNDMASK_SGPR) { SIND8("v_cndmask_b32_e64", ", s[20:21]"); }
        if (K == FMA_SGPR) { asm volatile(REP8("v_fma_f32 %0, %0, s20, %8\n v_fma_f32 %1, %1, s20, %8\n v_fma_f32 %2, %2, s20, %8\n v_fma_f32 %3, %3, s20, %8\n v_fma_f32 %4, %4, s20, %8\n v_fma_f32 %5, %5, s20, %8\n v_fma_f32 %6, %6, s20, %8\n v_fma_f32 %7, %7, s20, %8\n")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s20"); }
        if (K == MUL_SGPR) { asm volatile(REP8("v_mul_f32_e32 %0, s20, %0\n v_mul_f32_e32 %1, s20, %1\n v_mul_f32_e32 %2, s20, %2\n v_mul_f32_e32 %3, s20, %3\n v_mul_f32_e32 %4, s20, %4\n v_mul_f32_e32 %5, s20, %5\n v_mul_f32_e32 %6, s20, %6\n v_mul_f32_e32 %7, s20, %7\n")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20"); }
        if (K == CMP_VCC) { asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %1, %2\n v_cmp_lt_f32_e32 vcc, %2, %3\n v_cmp_lt_f32_e32 vcc, %3, %4\n v_cmp_lt_f32_e32 vcc, %4, %5\n v_cmp_lt_f32_e32 vcc, %5, %6\n v_cmp_lt_f32_e32 vcc, %6, %7\n v_cmp_lt_f32_e32 vcc, %7, %0\n")
            :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc"); }
        if (K == CMP_SGPR) { asm volatile(REP8("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cmp_lt_f32_e64 s[24:25], %2, %3\n v_cmp_lt_f32_e64 s[26:27], %3, %4\n v_cmp_lt_f32_e64 s[20:21], %4, %5\n v_cmp_lt_f32_e64 s[22:23], %5, %6\n v_cmp_lt_f32_e64 s[24:25], %6, %7\n v_cmp_lt_f32_e64 s[26:27], %7, %0\n")
            :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); }
        if (K == FMAAK) { asm volatile(REP8("v_fmaak_f32 %0, %0, %8, 0x3a83126f\n v_fmaak_f32 %1, %1, %8, 0x3a83126f\n v_fmaak_f32 %2, %2, %8, 0x3a83126f\n v_fmaak_f32 %3, %3, %8, 0x3a83126f\n v_fmaak_f32 %4, %4, %8, 0x3a83126f\n v_fmaak_f32 %5, %5, %8, 0x3a83126f\n v_fmaak_f32 %6, %6, %8, 0x3a83126f\n v_fmaak_f32 %7, %7, %8, 0x3a83126f\n")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        if (K == MOV_IND) { asm volatile(REP8("v_mov_b32_e32 %0, %1\n v_mov_b32_e32 %1, %2\n v_mov_b32_e32 %2, %3\n v_mov_b32_e32 %3, %4\n v_mov_b32_e32 %4, %5\n v_mov_b32_e32 %5, %6\n v_mov_b32_e32 %6, %7\n v_mov_b32_e32 %7, %0\n")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (K == FMA_CND_MIX) { asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_fma_f32 %2, %2, %8, %9\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_fma_f32 %4, %4, %8, %9\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_fma_f32 %6, %6, %8, %9\n v_cndmask_b32_e32 %7, %7, %8, vcc\n")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p1.y;
}

static int instr_per_rep(int k) {
    switch (k) { case LDS_RT: return 8; case BPERMUTE_DEP: return 8; case LDS_RD128: return 8; case BARRIER: return 8; case FMA_2CH: case PKFMA_2CH: return 128; case FMA_4CH: return 256; default: return 64; }
}

template <int K>
static void run(int blocks, int reps, float* out, unsigned long long* cyc, FILE* js, bool& first) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_bench<K>), dim3(blocks), dim3(64), 0, 0, out, cyc, reps, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int t = 0; t < 5; ++t) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_bench<K>), dim3(blocks), dim3(64), 0, 0, out, cyc, reps, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (auto v : h) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
    const double n = (double)reps * instr_per_rep(K);
    const double wave_cpi = sum / blocks / n;
    const double f_ghz = mx / (best * 1e6);   // the slowest wave spans (almost) the whole launch: shader cycles per wall ns
    const double simd_cpi = best * 1e6 * f_ghz * 1024.0 / ((double)blocks * n);
    printf("%-52s waves %5d (%5.2f / SIMD)  wave cyc/instr %7.2f  wall %8.1f us  f~%.2f GHz  SIMD cyc/instr %6.2f\n", kind_name[K], blocks, blocks / 1024.0, wave_cpi,
           best * 1e3, f_ghz, blocks >= 1024 ? simd_cpi : 0.0);
    fprintf(js, "%s\n  {\"kind\": \"%s\", \"wavefronts\": %d, \"per_simd\": %.3f, \"wave_cycles_per_instr\": %.3f, \"wall_us\": %.2f, \"shader_ghz_est\": %.3f, \"simd_cycles_per_instr\": %.3f}",
            first ? "" : ",", kind_name[K], blocks, blocks / 1024.0, wave_cpi, best * 1e3, f_ghz, blocks >= 1024 ? simd_cpi : 0.0);
    first = false;
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int K>
static void sweep(float* out, unsigned long long* cyc, FILE* js, bool& first) {
    const int reps = (K == LDS_RT || K == BPERMUTE_DEP || K == LDS_RD128 || K == BARRIER) ? 500 : 400;
    for (int blocks : {1, 256, 1024, 2048, 4096, 8192}) run<K>(blocks, reps, out, cyc, js, first);
}

template <int K>
static void sweep_all(float* out, unsigned long long* cyc, FILE* js, bool& first) {
    sweep<K>(out, cyc, js, first);
    if constexpr (K + 1 < NKINDS) sweep_all<K + 1>(out, cyc, js, first);
}

int main(int argc, char** argv) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 8192 * 64 * sizeof(float));
    hipMalloc(&cyc, 8192 * sizeof(unsigned long long));
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("# %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    FILE* js = fopen(argc > 1 ? argv[1] : "valu_issue.json", "w");
    fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"results\": [", p.gcnArchName, p.multiProcessorCount);
    bool first = true;
    sweep_all<0>(out, cyc, js, first);
    fprintf(js, "\n]}\n");
    fclose(js);
    return 0;
}
