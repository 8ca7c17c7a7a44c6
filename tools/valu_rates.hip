// tools/valu_rates.hip — measured issue cost of the vector instructions the frame kernels are made of
// (gfx950). Feeds the cost column of profiles/*_isa_budget.md: one wavefront's stream of 8 independent
// chains of ONE opcode, W wavefronts per SIMD (1, 2, 4, 8), timed with s_memtime inside the kernel.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates
//
// Output: JSON {op: {"w1": cycles per instruction per wavefront, "w2": ..., "simd_w8": cycles per
// instruction per SIMD at 8 wavefronts}}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define REP 4096 // loop trips: ~65k instructions per wavefront (>= 0.1 ms), so that the W workgroups of a CU overlap fully
#define UNR 16   // instructions per trip (2 per chain)

#define CHAIN8(OPSTR)                                                        \
    asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7) \
                 OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7) \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c) : "vcc", "s10", "s11")

#define OP_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define OP_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define OP_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define OP_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define OP_DIVFIX(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %9\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define OP_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define OP_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define OP_CVTUF(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define OP_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define OP_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define OP_CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define OP_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %9\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMPCNDS(i) "v_cmp_lt_f32 s[10:11], %" #i ", %9\nv_cndmask_b32 %" #i ", %" #i ", %8, s[10:11]\n"
#define OP_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[10:11]\n"
#define OP_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define OP_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define OP_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define OP_SUB(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define OP_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define OP_FMANEG(i) "v_fma_f32 %" #i ", -%" #i ", %8, %9\n"
#define OP_MULABS(i) "v_mul_f32 %" #i ", |%" #i "|, %8\n"
#define OP_CMPS(i) "v_cmp_lt_f32 s[10:11], %" #i ", %8\n"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_SUBREV(i) "v_subrev_u32 %" #i ", %" #i ", %8\n"
#define OP_ASHR(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define OP_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %8\n"
#define OP_FREXPM(i) "v_frexp_mant_f32 %" #i ", %" #i "\n"
#define OP_RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float* out, uint64_t* cycles, float b, float c)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + 0.001f * (threadIdx.x + i);
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r)
    {
        if (OP == 0) CHAIN8(OP_FMA);
        if (OP == 1) CHAIN8(OP_MUL);
        if (OP == 2) CHAIN8(OP_ADD);
        if (OP == 3) CHAIN8(OP_ADDU);
        if (OP == 4) CHAIN8(OP_MULLO);
        if (OP == 5) CHAIN8(OP_MULHI);
        if (OP == 6) CHAIN8(OP_MUL24);
        if (OP == 7) CHAIN8(OP_RCP);
        if (OP == 8) CHAIN8(OP_SQRT);
        if (OP == 9) CHAIN8(OP_RSQ);
        if (OP == 10) CHAIN8(OP_LOG);
        if (OP == 11) CHAIN8(OP_EXP);
        if (OP == 12) CHAIN8(OP_DIVFIX);
        if (OP == 13) CHAIN8(OP_DIVFMAS);
        if (OP == 14) CHAIN8(OP_DIVSCALE);
        if (OP == 15) CHAIN8(OP_CNDMASK);
        if (OP == 16) CHAIN8(OP_ALIGNBIT);
        if (OP == 17) CHAIN8(OP_LSHR);
        if (OP == 18) CHAIN8(OP_XOR);
        if (OP == 19) CHAIN8(OP_ADD3);
        if (OP == 20) CHAIN8(OP_CVTFU);
        if (OP == 21) CHAIN8(OP_CVTUF);
        if (OP == 22) CHAIN8(OP_MAX);
        if (OP == 23) CHAIN8(OP_MAX3);
        if (OP == 24) CHAIN8(OP_CMP);
        if (OP == 25) CHAIN8(OP_BFE);
        if (OP == 26) CHAIN8(OP_CVTUB);
        if (OP == 27) CHAIN8(OP_CNDS);
        if (OP == 28) CHAIN8(OP_MIN);
        if (OP == 29) CHAIN8(OP_MED3);
        if (OP == 30) CHAIN8(OP_AND);
        if (OP == 31) CHAIN8(OP_OR);
        if (OP == 32) CHAIN8(OP_SUB);
        if (OP == 33) CHAIN8(OP_FMAC);
        if (OP == 34) CHAIN8(OP_PERM);
        if (OP == 35) CHAIN8(OP_LSHLOR);
        if (OP == 36) CHAIN8(OP_MOV);
        if (OP == 37) CHAIN8(OP_FMANEG);
        if (OP == 38) CHAIN8(OP_MULABS);
        if (OP == 39) CHAIN8(OP_CMPS);
        if (OP == 40) CHAIN8(OP_MAD24);
        if (OP == 41) CHAIN8(OP_SUBREV);
        if (OP == 42) CHAIN8(OP_ASHR);
        if (OP == 43) CHAIN8(OP_LDEXP);
        if (OP == 44) CHAIN8(OP_FREXPM);
        if (OP == 45) CHAIN8(OP_RNDNE);
        if (OP == 46) CHAIN8(OP_CND64);
        if (OP == 47) CHAIN8(OP_CMPCND);
        if (OP == 48) CHAIN8(OP_CMPCNDS);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// 64-bit ops and packed f32 need register pairs: separate kernels
__global__ __launch_bounds__(256) void k_rate_mad64(float* out, uint64_t* cycles, uint32_t b, uint32_t c)
{
    uint64_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = 0x123456789ull * (threadIdx.x + i + 1);
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r)
    {
#define OP_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
        asm volatile(OP_MAD64(0) OP_MAD64(1) OP_MAD64(2) OP_MAD64(3) OP_MAD64(4) OP_MAD64(5) OP_MAD64(6) OP_MAD64(7)
                     OP_MAD64(0) OP_MAD64(1) OP_MAD64(2) OP_MAD64(3) OP_MAD64(4) OP_MAD64(5) OP_MAD64(6) OP_MAD64(7)
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "v"(c) : "vcc");
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k_rate_pk(float* out, uint64_t* cycles, float b0, float c0)
{
    float2v a[8];
    float2v b = {b0, b0}, c = {c0, c0};
    for (int i = 0; i < 8; ++i) a[i] = float2v{1.0f + 0.001f * (threadIdx.x + i), 1.0f};
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r)
    {
#define OP_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
        if (OP == 0)
            asm volatile(OP_PKFMA(0) OP_PKFMA(1) OP_PKFMA(2) OP_PKFMA(3) OP_PKFMA(4) OP_PKFMA(5) OP_PKFMA(6) OP_PKFMA(7)
                         OP_PKFMA(0) OP_PKFMA(1) OP_PKFMA(2) OP_PKFMA(3) OP_PKFMA(4) OP_PKFMA(5) OP_PKFMA(6) OP_PKFMA(7)
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(b), "v"(c));
        else
            asm volatile(OP_PKMUL(0) OP_PKMUL(1) OP_PKMUL(2) OP_PKMUL(3) OP_PKMUL(4) OP_PKMUL(5) OP_PKMUL(6) OP_PKMUL(7)
                         OP_PKMUL(0) OP_PKMUL(1) OP_PKMUL(2) OP_PKMUL(3) OP_PKMUL(4) OP_PKMUL(5) OP_PKMUL(6) OP_PKMUL(7)
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(b), "v"(c));
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// binary64 (the portable sin/cos reduce and evaluate in double)
template <int OP>
__global__ __launch_bounds__(256) void k_rate_f64(float* out, uint64_t* cycles, double b, double c)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r)
    {
#define OP_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define OP_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define OP_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define F64CHAIN(OPSTR)                                                                                        \
    asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                      \
                 OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                      \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c))
        if (OP == 0) F64CHAIN(OP_FMA64);
        if (OP == 1) F64CHAIN(OP_MUL64);
        if (OP == 2) F64CHAIN(OP_ADD64);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// conversions between binary32 and binary64: a f32 -> f64 -> f32 round trip per chain step (2 instructions)
__global__ __launch_bounds__(256) void k_rate_cvt64(float* out, uint64_t* cycles)
{
    float a[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 0.001f * (threadIdx.x + i); d[i] = 0.0; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r)
    {
#define OP_CVTRT(i, j) "v_cvt_f64_f32 %" #j ", %" #i "\nv_cvt_f32_f64 %" #i ", %" #j "\n"
        asm volatile(OP_CVTRT(0, 8) OP_CVTRT(1, 9) OP_CVTRT(2, 10) OP_CVTRT(3, 11) OP_CVTRT(4, 12) OP_CVTRT(5, 13) OP_CVTRT(6, 14) OP_CVTRT(7, 15)
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                       "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* NAMES[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24",
                              "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32", "v_exp_f32", "v_div_fixup_f32", "v_div_fmas_f32",
                              "v_div_scale_f32", "v_cndmask_b32", "v_alignbit_b32", "v_lshrrev_b32", "v_xor_b32", "v_add3_u32",
                              "v_cvt_f32_u32", "v_cvt_u32_f32", "v_max_f32", "v_max3_f32", "v_cmp_lt_f32", "v_bfe_u32", "v_cvt_f32_ubyte1",
                              "v_cndmask_b32 (sgpr mask)", "v_min_f32", "v_med3_f32", "v_and_b32", "v_or_b32", "v_sub_f32", "v_fmac_f32",
                              "v_perm_b32", "v_lshl_or_b32", "v_mov_b32", "v_fma_f32 (neg mod)", "v_mul_f32 (abs mod)",
                              "v_cmp_lt_f32 (sgpr dst)", "v_mad_u32_u24", "v_subrev_u32", "v_ashrrev_i32", "v_ldexp_f32",
                              "v_frexp_mant_f32", "v_rndne_f32", "v_cndmask_b32_e64 (vcc)", "v_cmp+v_cndmask via vcc (per pair)",
                              "v_cmp+v_cndmask via sgpr pair (per pair)"};

template <int OP> static void launch(int grid, float* out, uint64_t* cyc) { k_rate<OP><<<grid, 256>>>(out, cyc, 1.0001f, 0.5f); }
typedef void (*launch_fn)(int, float*, uint64_t*);

int main()
{
    const int CUS = 256;
    float* out; uint64_t* cyc;
    hipMalloc(&out, (size_t)CUS * 8 * 256 * 4);
    hipMalloc(&cyc, (size_t)CUS * 8 * 4 * 8);
    std::vector<uint64_t> h((size_t)CUS * 8 * 4);
    launch_fn fns[] = {launch<0>, launch<1>, launch<2>, launch<3>, launch<4>, launch<5>, launch<6>, launch<7>, launch<8>, launch<9>,
                       launch<10>, launch<11>, launch<12>, launch<13>, launch<14>, launch<15>, launch<16>, launch<17>, launch<18>,
                       launch<19>, launch<20>, launch<21>, launch<22>, launch<23>, launch<24>, launch<25>, launch<26>, launch<27>,
                       launch<28>, launch<29>, launch<30>, launch<31>, launch<32>, launch<33>, launch<34>, launch<35>, launch<36>,
                       launch<37>, launch<38>, launch<39>, launch<40>, launch<41>, launch<42>, launch<43>, launch<44>, launch<45>,
                       launch<46>, launch<47>, launch<48>};
    const int nfn = sizeof(fns) / sizeof(fns[0]);
    printf("{\n");
    auto measure = [&](const char* name, auto&& go, bool last) {
        printf(" \"%s\": {", name);
        const int Ws[] = {1, 2, 4, 8};
        for (int wi = 0; wi < 4; ++wi)
        {
            const int W = Ws[wi];
            const int grid = CUS * W; // 256-thread workgroups = one wavefront per SIMD each; W workgroups per CU
            go(grid);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            go(grid);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0.0f;
            hipEventElapsedTime(&ms, e0, e1);
            hipEventDestroy(e0); hipEventDestroy(e1);
            hipMemcpy(h.data(), cyc, (size_t)grid * 4 * 8, hipMemcpyDeviceToHost);
            std::vector<uint64_t> v(h.begin(), h.begin() + (size_t)grid * 4);
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2];
            const double per_wave = med / (double)(REP * UNR);
            printf("\"w%d\": %.2f, ", W, per_wave);
            if (W == 8) printf("\"w8_wall_ns_per_instr_per_simd\": %.3f", (double)ms * 1e6 / ((double)REP * UNR * 8.0));
        }
        printf("}%s\n", last ? "" : ",");
    };
    for (int f = 0; f < nfn; ++f) measure(NAMES[f], [&](int g) { fns[f](g, out, cyc); }, false);
    measure("v_mad_u64_u32", [&](int g) { k_rate_mad64<<<g, 256>>>(out, cyc, 12345u, 6789u); }, false);
    measure("v_pk_fma_f32", [&](int g) { k_rate_pk<0><<<g, 256>>>(out, cyc, 1.0001f, 0.5f); }, false);
    measure("v_pk_mul_f32", [&](int g) { k_rate_pk<1><<<g, 256>>>(out, cyc, 1.0001f, 0.5f); }, false);
    measure("v_fma_f64", [&](int g) { k_rate_f64<0><<<g, 256>>>(out, cyc, 1.0001, 0.5); }, false);
    measure("v_mul_f64", [&](int g) { k_rate_f64<1><<<g, 256>>>(out, cyc, 1.0001, 0.5); }, false);
    measure("v_add_f64", [&](int g) { k_rate_f64<2><<<g, 256>>>(out, cyc, 1.0001, 0.5); }, false);
    measure("v_cvt_f64_f32 + v_cvt_f32_f64 (per instruction)", [&](int g) { k_rate_cvt64<<<g, 256>>>(out, cyc); }, true);
    printf("}\n");
    return 0;
}
