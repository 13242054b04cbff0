// issue_calib.hip -- what do the SQ counters count per instruction, and what does an instruction of a kind cost a SIMD?  (tools only)
// One kernel per instruction kind: every wave runs `iters` rounds of 64 independent instructions of that kind (eight destination
// registers in rotation, no dependency between neighbours), on 1 or 8 waves per SIMD.  Run each variant under
//   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
// and divide (tools/issue_calib.sh): counter / instructions, SIMD-cycles / instructions.
//   build: hipcc --offload-arch=gfx950 -O2 tools/diag/issue_calib.hip -o gpurun_out/issue_calib
//   run:   issue_calib <kind> <waves_per_simd> [iters]      kinds: add cndmask cmp cvt fma rcp fma64 salu nop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define R64(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op)
#define K_ADD(i) "v_add_u32 %" #i ", %8, %" #i "\n\t"
#define K_CND(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9\n\t"
#define K_CMP(i) "v_cmp_lt_f32_e64 %10, %" #i ", %8\n\t"
#define K_CVT(i) "v_cvt_f32_i32_e32 %" #i ", %" #i "\n\t"
#define K_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n\t"
#define K_RCP(i) "v_rcp_f32_e32 %" #i ", %" #i "\n\t"
#define K_SALU(i) "s_add_u32 %11, %11, 1\n\t"
#define K_NOP(i) "s_nop 0\n\t"

template <int KIND>
__global__ void __launch_bounds__(1024) calib(int iters, float* out)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, k = 1.0009765625f;
    unsigned long long m = 0x5555555555555555ull, mo; int sc = 0;
    double d0 = r0, d1 = r1, d2 = r2, d3 = r3;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(R64(K_ADD) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "s"(m));
        if (KIND == 1) asm volatile(R64(K_CND) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "s"(m));
        if (KIND == 2) asm volatile(R64(K_CMP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "s"(m), "s"(mo) : "memory");
        if (KIND == 3) asm volatile(R64(K_CVT) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "s"(m));
        if (KIND == 4) asm volatile(R64(K_FMA) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "s"(m));
        if (KIND == 5) asm volatile(R64(K_RCP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "s"(m));
        if (KIND == 6) {
            #pragma unroll
            for (int q = 0; q < 16; ++q) { d0 = __builtin_fma(d0, 1.0000001, 0.5); d1 = __builtin_fma(d1, 1.0000001, 0.5); d2 = __builtin_fma(d2, 1.0000001, 0.5); d3 = __builtin_fma(d3, 1.0000001, 0.5); }
        }
        if (KIND == 7) asm volatile(R64(K_SALU) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "s"(m), "s"(mo), "s"(sc) : "scc");
        if (KIND == 8) asm volatile(R64(K_NOP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "s"(m));
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(d0 + d1 + d2 + d3) == 12345.678f) out[0] = r0;      // keep the registers alive
}

int main(int argc, char** argv)
{
    const char* kinds[] = { "add", "cndmask", "cmp", "cvt", "fma", "rcp", "fma64", "salu", "nop" };
    if (argc < 3) { fprintf(stderr, "usage: issue_calib <kind> <waves_per_simd> [iters]\n"); return 2; }
    int kind = -1; for (int i = 0; i < 9; ++i) if (!strcmp(argv[1], kinds[i])) kind = i;
    const int wps = atoi(argv[2]), iters = argc > 3 ? atoi(argv[3]) : 2000;
    if (kind < 0 || (wps != 1 && wps != 2 && wps != 4 && wps != 8)) return 2;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    // wps waves per SIMD = 4 * wps waves per CU: one workgroup of 256 * wps threads per CU (wps = 8: two workgroups of 1024 threads)
    const int threads = wps == 8 ? 1024 : 256 * wps, blocks = wps == 8 ? 2 * cus : cus;
    float* out; hipMalloc(&out, 64);
    void (*k[])(int, float*) = { calib<0>, calib<1>, calib<2>, calib<3>, calib<4>, calib<5>, calib<6>, calib<7>, calib<8> };
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k[kind], dim3(blocks), dim3(threads), 0, 0, 10, out);      // warm-up
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k[kind], dim3(blocks), dim3(threads), 0, 0, iters, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double waves = (double)blocks * threads / 64, insts = waves * iters * 64;
    printf("calib %s waves_per_simd %d cus %d waves %.0f instructions_of_the_kind %.6g kernel_ms %.4f\n", kinds[kind], wps, cus, waves, insts, ms);
    return 0;
}
