// Issue cost (cycles per wave64 instruction per SIMD) of the FP64 / select instructions the hydro sweeps are made of.
// Each kernel runs REP x 8 independent chains of one instruction per lane; 1024 blocks x 256 threads (4 waves per SIMD on 256 CUs).
// build: hipcc -O3 --offload-arch=gfx950 valu_f64.hip -o valu_f64 ; run: ./valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int REP = 4096;

#define KERNEL(NAME, BODY)                                                                                   \
	__global__ void __launch_bounds__(256) NAME(double *out, double s)                                      \
	{                                                                                                        \
		double a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
		double b = s * 0.999, c = s * 1.001;                                                              \
		for (int r = 0; r < REP; ++r) {                                                                  \
			BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7)                      \
		}                                                                                                \
		out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                     \
	}

#define B_FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_MIN(x) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_MAX(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define B_RSQ(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
#define B_SQRT(x) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
#define B_DIVFIX(x) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_DIVSCALE(x) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(b) : "vcc");
#define B_DIVFMAS(x) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define B_CMP(x) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define B_MOV64(x) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(b));
#define B_PKFMA32(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_DIVC(x) x = b / x;
#define B_SQRTC(x) x = sqrt(x);
#define B_MINC(x) x = (b < x) ? b : x;
#define B_FMINC(x) x = fmin(x, b);
#define B_CMPCLASS(x) asm volatile("v_cmp_ge_f64 vcc, |%0|, %1" : : "v"(x), "v"(b) : "vcc");

KERNEL(k_fma, B_FMA) KERNEL(k_mul, B_MUL) KERNEL(k_add, B_ADD) KERNEL(k_min, B_MIN) KERNEL(k_max, B_MAX) KERNEL(k_rcp, B_RCP) KERNEL(k_rsq, B_RSQ)
KERNEL(k_sqrt, B_SQRT) KERNEL(k_divfix, B_DIVFIX) KERNEL(k_divscale, B_DIVSCALE) KERNEL(k_divfmas, B_DIVFMAS) KERNEL(k_cmp, B_CMP)
KERNEL(k_mov64, B_MOV64) KERNEL(k_pkfma32, B_PKFMA32)
KERNEL(k_divc, B_DIVC) KERNEL(k_sqrtc, B_SQRTC) KERNEL(k_minc, B_MINC) KERNEL(k_fminc, B_FMINC) KERNEL(k_cmpabs, B_CMPCLASS)


#define KERNEL32(NAME, BODY)                                                                                 \
	__global__ void __launch_bounds__(256) NAME(double *out, double s)                                      \
	{                                                                                                        \
		float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
		float b = s * 0.999, c = s * 1.001;                                                               \
		for (int r = 0; r < REP; ++r) {                                                                  \
			BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7)                      \
		}                                                                                                \
		out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                     \
	}
#define C_CND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define C_MOV32(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(b));
#define C_FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define C_AND(x) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(x));
KERNEL32(k_cnd, C_CND) KERNEL32(k_mov32, C_MOV32) KERNEL32(k_fma32, C_FMA32) KERNEL32(k_fabs, C_AND)

int main(int argc, char **argv)
{
	const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
	const int nblocks = 256 * waves_per_simd; // 256 CUs, one 256-thread block = 1 wave per SIMD of a CU
	double *out;
	CHK(hipMalloc(&out, sizeof(double) * nblocks * 256));
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0));
	CHK(hipEventCreate(&e1));
	struct K { const char *name; void (*fn)(double *, double); int per; };
	std::vector<K> ks = {{"v_fma_f64", k_fma, 1}, {"v_mul_f64", k_mul, 1}, {"v_add_f64", k_add, 1}, {"v_min_f64", k_min, 1}, {"v_max_f64", k_max, 1},
			     {"v_rcp_f64", k_rcp, 1}, {"v_rsq_f64", k_rsq, 1}, {"v_sqrt_f64", k_sqrt, 1}, {"v_div_fixup_f64", k_divfix, 1},
			     {"v_div_scale_f64", k_divscale, 1}, {"v_div_fmas_f64", k_divfmas, 1}, {"v_cmp_lt_f64", k_cmp, 1},
			     {"v_cndmask_b32", k_cnd, 1}, {"v_mov_b64", k_mov64, 1}, {"v_mov_b32", k_mov32, 1},
			     {"v_fma_f32", k_fma32, 1}, {"v_pk_fma_f32", k_pkfma32, 1}, {"C: b / x", k_divc, 1}, {"C: sqrt(x)", k_sqrtc, 1},
			     {"C: (b<x)?b:x", k_minc, 1}, {"C: fmin(x,b)", k_fminc, 1}, {"v_and_b32 (fabs)", k_fabs, 1}, {"v_cmp_ge_f64 |x|", k_cmpabs, 1}};
	// clock estimate from v_fma_f64 is circular; report ns per wave-instruction per SIMD and cycles at the clock the v_fma_f32 row implies if it is 2 cycles... print both raw
	printf("%-28s %12s %14s\n", "instruction", "ms", "ns/instr/SIMD");
	for (auto &k : ks) {
		hipLaunchKernelGGL(k.fn, dim3(nblocks), dim3(256), 0, 0, out, 1.0000001);
		CHK(hipDeviceSynchronize());
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL(k.fn, dim3(nblocks), dim3(256), 0, 0, out, 1.0000001);
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		const double instr_per_simd = static_cast<double>(REP) * 8 * waves_per_simd; // wave-instructions issued on one SIMD
		printf("%-28s %12.4f %14.3f\n", k.name, ms, ms * 1e6 / instr_per_simd);
	}
	return 0;
}
