// Cost of FP64 selects on gfx950: v_cmp + 2 x v_cndmask_b32 (what `(b < a) ? b : a` compiles to) against v_min_f64, with hard-coded registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int REP = 4096;
#define X8(M) M(10,11) M(12,13) M(14,15) M(16,17) M(18,19) M(20,21) M(22,23) M(24,25)
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v30","v31","vcc","s10","s11"
#define KERNEL(NAME, PRE, M)                                                                    \
	__global__ void __launch_bounds__(256) NAME(double *out, double s)                          \
	{                                                                                           \
		asm volatile(PRE ::: CLOB);                                                             \
		for (int r = 0; r < REP; ++r) {                                                         \
			asm volatile(X8(M) ::: CLOB);                                                       \
		}                                                                                       \
		double res;                                                                             \
		asm volatile("v_mov_b64 %0, v[10:11]" : "=v"(res) :: CLOB);                             \
		out[blockIdx.x * 256 + threadIdx.x] = res;                                              \
	}
#define INIT "v_mov_b32 v30, 0x1234\n v_mov_b32 v31, 0x3ff00000\n v_mov_b32 v10, 1\n v_mov_b32 v11, 0x3ff10000\n v_mov_b32 v12, 1\n v_mov_b32 v13, 0x3ff10000\n v_mov_b32 v14, 1\n v_mov_b32 v15, 0x3ff10000\n v_mov_b32 v16, 1\n v_mov_b32 v17, 0x3ff10000\n v_mov_b32 v18, 1\n v_mov_b32 v19, 0x3ff10000\n v_mov_b32 v20, 1\n v_mov_b32 v21, 0x3ff10000\n v_mov_b32 v22, 1\n v_mov_b32 v23, 0x3ff10000\n v_mov_b32 v24, 1\n v_mov_b32 v25, 0x3ff10000\n v_cmp_lt_f64 vcc, v[30:31], v[10:11]\n s_mov_b64 s[10:11], vcc\n s_nop 4\n"
#define STR(x) #x
#define M_MINSEL(a,b) "v_cmp_lt_f64 vcc, v[30:31], v[" STR(a) ":" STR(b) "]\n s_nop 1\n v_cndmask_b32 v" STR(a) ", v" STR(a) ", v30, vcc\n v_cndmask_b32 v" STR(b) ", v" STR(b) ", v31, vcc\n"
#define M_MINSEL_S(a,b) "v_cmp_lt_f64 s[10:11], v[30:31], v[" STR(a) ":" STR(b) "]\n s_nop 1\n v_cndmask_b32 v" STR(a) ", v" STR(a) ", v30, s[10:11]\n v_cndmask_b32 v" STR(b) ", v" STR(b) ", v31, s[10:11]\n"
#define M_MIN(a,b) "v_min_f64 v[" STR(a) ":" STR(b) "], v[" STR(a) ":" STR(b) "], v[30:31]\n"
#define M_CND_VCC(a,b) "v_cndmask_b32 v" STR(a) ", v" STR(a) ", v30, vcc\n v_cndmask_b32 v" STR(b) ", v" STR(b) ", v31, vcc\n"
#define M_CND_S(a,b) "v_cndmask_b32 v" STR(a) ", v" STR(a) ", v30, s[10:11]\n v_cndmask_b32 v" STR(b) ", v" STR(b) ", v31, s[10:11]\n"
#define M_CND_E64VCC(a,b) "v_cndmask_b32_e64 v" STR(a) ", v" STR(a) ", v30, vcc\n v_cndmask_b32_e64 v" STR(b) ", v" STR(b) ", v31, vcc\n"
#define M_CND_E32(a,b) "v_cndmask_b32_e32 v" STR(a) ", v" STR(a) ", v30, vcc\n v_cndmask_b32_e32 v" STR(b) ", v" STR(b) ", v31, vcc\n"
#define M_CND_E32_DIFF(a,b) "v_cndmask_b32_e32 v" STR(a) ", v31, v30, vcc\n v_cndmask_b32_e32 v" STR(b) ", v30, v31, vcc\n"
#define M_CND_S_DIFF(a,b) "v_cndmask_b32_e64 v" STR(a) ", v31, v30, s[10:11]\n v_cndmask_b32_e64 v" STR(b) ", v30, v31, s[10:11]\n"
#define M_CMP(a,b) "v_cmp_lt_f64 vcc, v[30:31], v[" STR(a) ":" STR(b) "]\n"
#define M_CMP_NOP(a,b) "v_cmp_lt_f64 vcc, v[30:31], v[" STR(a) ":" STR(b) "]\n s_nop 1\n"
#define M_MOV2(a,b) "v_mov_b32 v" STR(a) ", v30\n v_mov_b32 v" STR(b) ", v31\n"
#define M_FMA(a,b) "v_fma_f64 v[" STR(a) ":" STR(b) "], v[" STR(a) ":" STR(b) "], v[30:31], v[30:31]\n"
#define M_MAXMIN(a,b) "v_max_f64 v[" STR(a) ":" STR(b) "], v[" STR(a) ":" STR(b) "], v[" STR(a) ":" STR(b) "]\n v_min_f64 v[" STR(a) ":" STR(b) "], v[" STR(a) ":" STR(b) "], v[30:31]\n"
#define M_BFI(a,b) "v_bfi_b32 v" STR(a) ", v30, v" STR(a) ", v31\n v_bfi_b32 v" STR(b) ", v30, v" STR(b) ", v31\n"
KERNEL(k_minsel, INIT, M_MINSEL) KERNEL(k_minsel_s, INIT, M_MINSEL_S) KERNEL(k_min, INIT, M_MIN) KERNEL(k_cnd_vcc, INIT, M_CND_VCC) KERNEL(k_cnd_s, INIT, M_CND_S)
KERNEL(k_e64vcc, INIT, M_CND_E64VCC) KERNEL(k_e32, INIT, M_CND_E32) KERNEL(k_e32d, INIT, M_CND_E32_DIFF) KERNEL(k_sd, INIT, M_CND_S_DIFF) KERNEL(k_cmp, INIT, M_CMP) KERNEL(k_cmp_nop, INIT, M_CMP_NOP) KERNEL(k_mov2, INIT, M_MOV2) KERNEL(k_fma, INIT, M_FMA) KERNEL(k_maxmin, INIT, M_MAXMIN) KERNEL(k_bfi, INIT, M_BFI)
int main(int argc, char **argv)
{
	const int w = argc > 1 ? atoi(argv[1]) : 4;
	double *out;
	if (hipMalloc(&out, sizeof(double) * 256 * w * 256) != hipSuccess) return 1;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	struct K { const char *name; void (*fn)(double *, double); };
	std::vector<K> ks = {{"v_fma_f64", k_fma}, {"v_min_f64", k_min}, {"cmp(vcc)+nop1+2cndmask", k_minsel}, {"cmp(sgpr)+nop1+2cndmask", k_minsel_s}, {"2x v_cndmask_b32 vcc", k_cnd_vcc},
			     {"2x v_cndmask_b32 sgpr", k_cnd_s}, {"2x cndmask_e64 vcc", k_e64vcc}, {"2x cndmask_e32 vcc", k_e32}, {"2x cndmask_e32 vcc (no dep)", k_e32d}, {"2x cndmask_e64 sgpr (no dep)", k_sd}, {"v_cmp_lt_f64", k_cmp}, {"v_cmp_lt_f64 + s_nop 1", k_cmp_nop}, {"2x v_mov_b32", k_mov2},
			     {"v_max(x,x)+v_min", k_maxmin}, {"2x v_bfi_b32", k_bfi}};
	printf("%-28s %12s %16s   (%d waves/SIMD)\n", "pattern", "ms", "ns/pattern/SIMD", w);
	for (auto &k : ks) {
		hipLaunchKernelGGL(k.fn, dim3(256 * w), dim3(256), 0, 0, out, 1.0);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		hipLaunchKernelGGL(k.fn, dim3(256 * w), dim3(256), 0, 0, out, 1.0);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("%-28s %12.4f %16.3f\n", k.name, ms, ms * 1e6 / (double(REP) * 8 * w));
	}
	return 0;
}
