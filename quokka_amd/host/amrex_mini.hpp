// amrex_mini.hpp — the slice of the AMReX API that the reference's hot-path callers and the three config problems use
// (Box, IntVect, Array4, GpuArray, BCRec, Geometry, MultiFab / iMultiFab with device storage, ParmParse, ParallelFor
// as HIP kernels, Print / Abort).  AMReX is an un-vendored submodule of the reference and is absent from this
// image, so problem generators written against the reference's surface are compiled against this header instead.
// Device data live in HIP allocations; the descriptor table of a MultiFab (`arrays()`) is what the C-ABI consumes
// (qk_array4 == amrex::Array4<Real>).
//
// One way to compile a problem file against it: as HIP (`-x hip`, what `make refproblem P=<dir>` uses to compile the reference's problem
// files UNCHANGED, read in place from /root/reference/src/problems): AMREX_GPU_DEVICE is __device__, ParallelFor launches the lambda as a HIP
// kernel over device arrays (MFIter / array(mfi) / const_array(mfi) / Gpu::DeviceVector / ReduceSum as the reference uses them); initial
// conditions, ErrorEst, setCustomBoundaryConditions and the analysis lambdas of a problem run on the GPU exactly as written.  This is
// problem-side glue (one backend: HIP); the hot path stays behind include/quokka_amd.h.
#ifndef QK_HOST_AMREX_MINI_HPP_
#define QK_HOST_AMREX_MINI_HPP_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cassert>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <optional>
#include <tuple>
#include <sstream>
#include <stdexcept>
#include <string>
#include <cctype>
#include <type_traits>
#include <utility>
#include <random>
#include <vector>

#include "quokka_amd.h"

#ifndef AMREX_SPACEDIM
#define AMREX_SPACEDIM 3
#endif
// BL_PROFILE("name") declares a timer object in AMReX; the problem files write it as a statement, some as `const BL_PROFILE(...)`
#define QK_CAT2(a, b) a##b
#define QK_CAT(a, b) QK_CAT2(a, b)
#define BL_PROFILE(name) int QK_CAT(qk_bl_profile_, __LINE__) = 0
#define AMREX_GPU_DEVICE __device__
#define AMREX_GPU_HOST_DEVICE __host__ __device__
#define AMREX_GPU_HOST __host__
#define QK_HD __host__ __device__
#define AMREX_USE_GPU 1
#define AMREX_USE_HIP 1
#define AMREX_GPU_MANAGED __managed__
#define AMREX_FORCE_INLINE inline
#define AMREX_INLINE inline
#define AMREX_NO_INLINE
#define AMREX_RESTRICT __restrict__
#define AMREX_ASSERT(x) ((void)0)
#define AMREX_ALWAYS_ASSERT(x)                                                                                                                       \
	do {                                                                                                                                         \
		if (!(x)) {                                                                                                                          \
			amrex::Abort("assertion failed: " #x);                                                                                       \
		}                                                                                                                                    \
	} while (0)
// (usable inside __host__ __device__ hooks that kernels now call — RadTophat's opacity: the message is passed as the literal it is, so that the
// device overload of amrex::Abort applies there)
#define AMREX_ALWAYS_ASSERT_WITH_MESSAGE(x, msg)                                                                                                     \
	do {                                                                                                                                         \
		if (!(x)) {                                                                                                                          \
			amrex::Abort(msg);                                                                                                           \
		}                                                                                                                                    \
	} while (0)
#define AMREX_ASSERT_WITH_MESSAGE(x, msg) ((void)0)
#if AMREX_SPACEDIM == 1
#define AMREX_D_DECL(a, b, c) a
#define AMREX_D_TERM(a, b, c) a
#elif AMREX_SPACEDIM == 2
#define AMREX_D_DECL(a, b, c) a, b
#define AMREX_D_TERM(a, b, c) a b
#else
#define AMREX_D_DECL(a, b, c) a, b, c
#define AMREX_D_TERM(a, b, c) a b c
#endif

#include "qk_comm.hpp"

namespace amrex
{
using Real = double;

using Long = long;
template <typename T> using Vector = std::vector<T>;
template <class T, std::size_t N> using Array = std::array<T, N>;

// inside kernels: stop the wave (the reference's device-side amrex::Abort traps as well)
__device__ inline void Abort(char const * /*msg*/) { __builtin_trap(); }
[[noreturn]] __host__ inline void Abort(char const *msg)
{
	std::fprintf(stderr, "amrex::Abort: %s\n", msg);
	std::exit(2);
}
[[noreturn]] inline void Abort(std::string const &msg)
{
	std::fprintf(stderr, "amrex::Abort: %s\n", msg.c_str());
	std::exit(2);
}
inline void ignore_unused(...) {}
// after every kernel launch of this header: a lambda that cannot be launched (too many registers, invalid configuration) must not go unnoticed
inline void qk_check_launch(char const *what)
{
	hipError_t const e = hipGetLastError();
	if (e != hipSuccess) {
		std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
		std::exit(2);
	}
}

struct PrintStream {
	template <typename T> auto operator<<(T const &v) -> PrintStream &
	{
		std::cout << v;
		return *this;
	}
	auto operator<<(std::ostream &(*f)(std::ostream &)) -> PrintStream &
	{
		std::cout << f;
		return *this;
	}
};
inline auto Print() -> PrintStream { return {}; }

struct Dim3 {
	int x, y, z;
};

template <typename T, int N> struct GpuArray {
	T arr[N > 0 ? N : 1];
	QK_HD constexpr auto operator[](int i) const -> T const & { return arr[i]; }
	QK_HD constexpr auto operator[](int i) -> T & { return arr[i]; }
	[[nodiscard]] QK_HD auto begin() const -> T const * { return arr; }
	[[nodiscard]] QK_HD auto end() const -> T const * { return arr + N; }
	[[nodiscard]] QK_HD auto data() const -> T const * { return arr; }
	[[nodiscard]] QK_HD auto data() -> T * { return arr; }
	[[nodiscard]] QK_HD static constexpr auto size() -> unsigned { return N; }
	QK_HD void fill(T const &v)
	{
		for (int i = 0; i < N; ++i) {
			arr[i] = v;
		}
	}
};

} // namespace amrex
namespace std
{
template <typename T, int N> struct tuple_size<amrex::GpuArray<T, N>> : integral_constant<size_t, static_cast<size_t>(N)> {
};
template <size_t I, typename T, int N> struct tuple_element<I, amrex::GpuArray<T, N>> {
	using type = T;
};
} // namespace std
namespace amrex
{
template <size_t I, typename T, int N> QK_HD auto get(GpuArray<T, N> const &a) -> T const & { return a.arr[I]; }
template <size_t I, typename T, int N> QK_HD auto get(GpuArray<T, N> &a) -> T & { return a.arr[I]; }
template <size_t I, typename T, int N> QK_HD auto get(GpuArray<T, N> &&a) -> T && { return static_cast<T &&>(a.arr[I]); }

struct IntVect {
	int v[3] = {0, 0, 0};
	IntVect() = default;
	QK_HD IntVect(int i, int j = 0, int k = 0) : v{i, j, k} {}
	QK_HD auto operator[](int d) const -> int { return v[d]; }
	QK_HD auto operator[](int d) -> int & { return v[d]; }
	[[nodiscard]] QK_HD auto dim3() const -> Dim3 { return {v[0], v[1], v[2]}; }
	[[nodiscard]] QK_HD auto toArray() const -> GpuArray<int, AMREX_SPACEDIM>
	{
		GpuArray<int, AMREX_SPACEDIM> a{};
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			a[d] = v[d];
		}
		return a;
	}
};


class Box
{
      public:
	int lo[3] = {0, 0, 0};
	int hi[3] = {0, 0, 0};
	Box() = default;
	Box(IntVect const &s, IntVect const &b)
	{
		for (int d = 0; d < 3; ++d) {
			lo[d] = s[d];
			hi[d] = b[d];
		}
	}
	[[nodiscard]] QK_HD auto smallEnd(int d) const -> int { return lo[d]; }
	[[nodiscard]] QK_HD auto bigEnd(int d) const -> int { return hi[d]; }
	[[nodiscard]] QK_HD auto smallEnd() const -> IntVect { return {lo[0], lo[1], lo[2]}; }
	[[nodiscard]] QK_HD auto bigEnd() const -> IntVect { return {hi[0], hi[1], hi[2]}; }
	[[nodiscard]] QK_HD auto length(int d) const -> int { return hi[d] - lo[d] + 1; }
	[[nodiscard]] QK_HD auto numPts() const -> Long { return static_cast<Long>(length(0)) * length(1) * length(2); }
	[[nodiscard]] QK_HD auto loVect3d() const -> GpuArray<int, 3> { return {{lo[0], lo[1], lo[2]}}; }
	[[nodiscard]] QK_HD auto hiVect3d() const -> GpuArray<int, 3> { return {{hi[0], hi[1], hi[2]}}; }
	[[nodiscard]] QK_HD auto contains(int i, int j, int k) const -> bool
	{
		return i >= lo[0] && i <= hi[0] && j >= lo[1] && j <= hi[1] && k >= lo[2] && k <= hi[2];
	}
	[[nodiscard]] QK_HD auto contains(IntVect const &c) const -> bool { return contains(c[0], c[1], c[2]); }
	[[nodiscard]] QK_HD auto ok() const -> bool { return hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2]; }
	// intersection (empty: ok() is false)
	[[nodiscard]] QK_HD auto operator&(Box const &o) const -> Box
	{
		Box r;
		for (int d = 0; d < 3; ++d) {
			r.lo[d] = lo[d] > o.lo[d] ? lo[d] : o.lo[d];
			r.hi[d] = hi[d] < o.hi[d] ? hi[d] : o.hi[d];
		}
		return r;
	}
};
inline auto makeSingleCellBox(int i, int j, int k) -> Box { return Box(IntVect(i, j, k), IntVect(i, j, k)); }
inline auto grow(Box b, int ng) -> Box
{
	for (int d = 0; d < AMREX_SPACEDIM; ++d) {
		b.lo[d] -= ng;
		b.hi[d] += ng;
	}
	return b;
}

// amrex::Array4<T> — layout shared with qk_array4 / qk_iarray4
template <typename T> struct Array4 {
	T *p = nullptr;
	Long jstride = 0, kstride = 0, nstride = 0;
	Dim3 begin{1, 1, 1};
	Dim3 end{0, 0, 0};
	int ncomp = 0;
	Array4() = default;
	QK_HD Array4(T *ptr, Box const &bx, int nc) : p(ptr), begin{bx.lo[0], bx.lo[1], bx.lo[2]}, end{bx.hi[0] + 1, bx.hi[1] + 1, bx.hi[2] + 1}, ncomp(nc)
	{
		jstride = bx.length(0);
		kstride = jstride * bx.length(1);
		nstride = kstride * bx.length(2);
	}
	// Array4<T const> from Array4<T>
	template <typename U, std::enable_if_t<std::is_same_v<std::remove_const_t<T>, U> && std::is_const_v<T>, int> = 0>
	QK_HD Array4(Array4<U> const &o) : p(o.p), jstride(o.jstride), kstride(o.kstride), nstride(o.nstride), begin(o.begin), end(o.end), ncomp(o.ncomp)
	{
	}
	QK_HD auto operator()(int i, int j, int k, int n = 0) const -> T & { return p[(i - begin.x) + jstride * (j - begin.y) + kstride * (k - begin.z) + nstride * n]; }
	QK_HD auto operator()(IntVect const &iv, int n = 0) const -> T & { return (*this)(iv[0], iv[1], iv[2], n); }
	[[nodiscard]] QK_HD auto nComp() const -> int { return ncomp; }
	[[nodiscard]] QK_HD auto contains(int i, int j, int k) const -> bool
	{
		return i >= begin.x && i < end.x && j >= begin.y && j < end.y && k >= begin.z && k < end.z;
	}
};
static_assert(sizeof(Array4<Real>) == sizeof(qk_array4), "amrex::Array4<Real> must match qk_array4");

// plain host loop over a box (diagnostics on staging copies; in host mode also what ParallelFor is)
template <typename F> void HostFor(Box const &bx, F &&f)
{
	for (int k = bx.lo[2]; k <= bx.hi[2]; ++k) {
		for (int j = bx.lo[1]; j <= bx.hi[1]; ++j) {
			for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) {
				f(i, j, k);
			}
		}
	}
}
// amrex::Loop: a serial loop over a box, callable inside kernels
template <typename F> QK_HD void Loop(Box const &bx, F const &f)
{
	for (int k = bx.lo[2]; k <= bx.hi[2]; ++k) {
		for (int j = bx.lo[1]; j <= bx.hi[1]; ++j) {
			for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) {
				f(i, j, k);
			}
		}
	}
}
// amrex::ParallelFor on the GPU: one thread per cell of the box, the lambda by value in the kernel arguments (default stream: ordered with
// the C-ABI calls of the driver, which use the same stream)
// Two mappings.  Rows of at least 32 cells: threads along x, one grid row per (j, k) — no index division at all, coalesced along x (a 64-bit
// division per thread, ~100 instructions, had made a one-store lambda like RadhydroShell's SetRadEnergySource cost 46 us per 128^3 box: 13 % of a
// coupled step of the shell).  Short rows (ghost slabs, 1-D pencils): the flat index, in 32-bit arithmetic whenever the box has fewer than 2^31 cells.
template <typename F> __global__ void qk_parfor_rows(Box bx, F f)
{
	const int i = bx.lo[0] + static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
	if (i <= bx.hi[0]) {
		f(i, bx.lo[1] + static_cast<int>(blockIdx.y), bx.lo[2] + static_cast<int>(blockIdx.z));
	}
}
template <typename F> __global__ void qk_parfor_rows_n(Box bx, F f)
{
	const int i = bx.lo[0] + static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
	const int nz = bx.length(2);
	const int comp = static_cast<int>(blockIdx.z) / nz;
	if (i <= bx.hi[0]) {
		f(i, bx.lo[1] + static_cast<int>(blockIdx.y), bx.lo[2] + (static_cast<int>(blockIdx.z) - comp * nz), comp);
	}
}
template <typename F> __global__ void qk_parfor_kernel(Box bx, F f)
{
	const Long n = static_cast<Long>(blockIdx.x) * blockDim.x + threadIdx.x;
	const Long npts = bx.numPts();
	if (n >= npts) {
		return;
	}
	const int nx = bx.length(0), ny = bx.length(1);
	int k, r;
	if (npts < (1LL << 31)) { // uniform
		const unsigned un = static_cast<unsigned>(n), nxy = static_cast<unsigned>(nx) * static_cast<unsigned>(ny);
		k = static_cast<int>(un / nxy);
		r = static_cast<int>(un - static_cast<unsigned>(k) * nxy);
	} else {
		k = static_cast<int>(n / (static_cast<Long>(nx) * ny));
		r = static_cast<int>(n - static_cast<Long>(k) * nx * ny);
	}
	const int j = r / nx;
	f(bx.lo[0] + (r - j * nx), bx.lo[1] + j, bx.lo[2] + k);
}
template <typename F> __global__ void qk_parfor_kernel_n(Box bx, int ncomp, F f)
{
	const Long n = static_cast<Long>(blockIdx.x) * blockDim.x + threadIdx.x;
	const Long npts = bx.numPts();
	if (n >= npts * ncomp) {
		return;
	}
	const int nx = bx.length(0), ny = bx.length(1);
	int comp, k, r;
	if (npts * ncomp < (1LL << 31)) { // uniform
		const unsigned un = static_cast<unsigned>(n), up = static_cast<unsigned>(npts), nxy = static_cast<unsigned>(nx) * static_cast<unsigned>(ny);
		comp = static_cast<int>(un / up);
		const unsigned cell = un - static_cast<unsigned>(comp) * up;
		k = static_cast<int>(cell / nxy);
		r = static_cast<int>(cell - static_cast<unsigned>(k) * nxy);
	} else {
		const Long cell = n % npts;
		comp = static_cast<int>(n / npts);
		k = static_cast<int>(cell / (static_cast<Long>(nx) * ny));
		r = static_cast<int>(cell - static_cast<Long>(k) * nx * ny);
	}
	const int j = r / nx;
	f(bx.lo[0] + (r - j * nx), bx.lo[1] + j, bx.lo[2] + k, comp);
}
inline auto qk_parfor_row_block(int nx) -> unsigned { return nx >= 256 ? 256U : static_cast<unsigned>((nx + 63) / 64 * 64); }

// Batched launches.  A problem hook like RadSystem<problem_t>::SetRadEnergySource is a HOST function that calls ParallelFor on ONE box; the driver
// calls it once per box, every call a launch of a few microseconds of work (RadhydroShell: 8 boxes x 2 calls x 10 substeps per step, 9.75 % of the
// step).  Inside a qk_parfor_batch_scope the row-mapped ParallelFor(Box, f) calls with the SAME lambda type are collected — box and closure by value,
// up to what the 4 KiB of kernel arguments hold — and leave as ONE launch (blockIdx.z = the call) when the scope ends, the pack is full or another
// kind of launch, a copy, an allocation or a synchronisation of the mirror arrives (QK_HOST_HIP, the other ParallelFor forms, launch, fills: each flushes
// first, so a hook that mixes launch kinds or synchronises and reads on the host sees its launches in order).  The calls of a pack must not depend on
// one another (one hook, one box each) and must not capture device buffers that die before the scope ends.
struct qk_parfor_batch_state {
	int depth = 0;
	void (*flush)() = nullptr;
};
inline auto qk_parfor_batch() -> qk_parfor_batch_state &
{
	static thread_local qk_parfor_batch_state s;
	return s;
}
inline void qk_parfor_batch_flush()
{
	auto &st = qk_parfor_batch();
	if (st.flush != nullptr) {
		auto *f = st.flush;
		st.flush = nullptr;
		f();
	}
}
struct qk_parfor_batch_scope {
	qk_parfor_batch_scope() { ++qk_parfor_batch().depth; }
	~qk_parfor_batch_scope()
	{
		if (--qk_parfor_batch().depth == 0) {
			qk_parfor_batch_flush();
		}
	}
	qk_parfor_batch_scope(qk_parfor_batch_scope const &) = delete;
	auto operator=(qk_parfor_batch_scope const &) -> qk_parfor_batch_scope & = delete;
};
template <typename F, int N> struct qk_parfor_pack {
	int n;
	Box bx[N];
	alignas(F) unsigned char f[N][sizeof(F)];
};
template <typename F, int N> __global__ void qk_parfor_rows_batch(qk_parfor_pack<F, N> p)
{
	const Box bx = p.bx[blockIdx.z];
	const int i = bx.lo[0] + static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
	const unsigned ny = static_cast<unsigned>(bx.length(1)), r = blockIdx.y;
	if (i > bx.hi[0] || r >= ny * static_cast<unsigned>(bx.length(2))) {
		return;
	}
	const unsigned k = r / ny;
	(*reinterpret_cast<F const *>(p.f[blockIdx.z]))(i, bx.lo[1] + static_cast<int>(r - k * ny), bx.lo[2] + static_cast<int>(k));
}
template <typename F> constexpr auto qk_parfor_pack_size() -> int
{
	constexpr size_t per = sizeof(Box) + ((sizeof(F) + alignof(F) - 1) / alignof(F)) * alignof(F);
	constexpr size_t n = (3584 - 64) / per; // (kernel arguments: 4 KiB, some of it the runtime's)
	return n >= 16 ? 16 : static_cast<int>(n);
}
template <typename F> auto qk_parfor_try_batch(Box const &bx, F const &f) -> bool
{
	constexpr int N = qk_parfor_pack_size<F>();
	if constexpr (N < 2 || !std::is_trivially_copyable_v<F>) {
		return false;
	} else {
		static thread_local qk_parfor_pack<F, N> pack{};
		auto launch = +[]() {
			if (pack.n == 0) {
				return;
			}
			unsigned nx = 0, rows = 0;
			for (int q = 0; q < pack.n; ++q) {
				nx = std::max(nx, static_cast<unsigned>(pack.bx[q].length(0)));
				rows = std::max(rows, static_cast<unsigned>(pack.bx[q].length(1)) * static_cast<unsigned>(pack.bx[q].length(2)));
			}
			const unsigned tb = qk_parfor_row_block(static_cast<int>(nx));
			hipLaunchKernelGGL((qk_parfor_rows_batch<F, N>), dim3((nx + tb - 1) / tb, rows, static_cast<unsigned>(pack.n)), dim3(tb), 0, nullptr, pack);
			pack.n = 0;
			qk_check_launch("amrex::ParallelFor (batched)");
		};
		auto &st = qk_parfor_batch();
		if (st.flush != nullptr && st.flush != launch) {
			qk_parfor_batch_flush(); // a pack of another lambda type is pending
		}
		if (static_cast<Long>(bx.length(1)) * bx.length(2) > 65535) { // (rows of the batch kernel live in gridDim.y)
			qk_parfor_batch_flush();
			return false;
		}
		pack.bx[pack.n] = bx;
		std::memcpy(pack.f[pack.n], &f, sizeof(F));
		++pack.n;
		st.flush = launch;
		if (pack.n == N) {
			qk_parfor_batch_flush();
		}
		return true;
	}
}

template <typename F> void ParallelFor(Box const &bx, F const &f)
{
	const Long n = bx.numPts();
	if (n <= 0) {
		return;
	}
	const int nx = bx.length(0);
	if (qk_parfor_batch().depth > 0) {
		if (nx >= 32 && qk_parfor_try_batch(bx, f)) {
			return;
		}
		qk_parfor_batch_flush(); // (keeps the order of the launches)
	}
	if (nx >= 32 && bx.length(1) <= 65535 && bx.length(2) <= 65535) {
		const unsigned tb = qk_parfor_row_block(nx);
		hipLaunchKernelGGL(qk_parfor_rows<F>, dim3((static_cast<unsigned>(nx) + tb - 1) / tb, static_cast<unsigned>(bx.length(1)), static_cast<unsigned>(bx.length(2))),
				   dim3(tb), 0, nullptr, bx, f);
	} else {
		hipLaunchKernelGGL(qk_parfor_kernel<F>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, bx, f);
	}
	qk_check_launch("amrex::ParallelFor");
}
template <typename F> void ParallelFor(Box const &bx, int ncomp, F const &f)
{
	const Long n = bx.numPts() * ncomp;
	if (n <= 0) {
		return;
	}
	qk_parfor_batch_flush(); // (keeps the order of the launches; only the three-index form is batched)
	const int nx = bx.length(0);
	if (nx >= 32 && bx.length(1) <= 65535 && static_cast<Long>(bx.length(2)) * ncomp <= 65535) {
		const unsigned tb = qk_parfor_row_block(nx);
		hipLaunchKernelGGL(qk_parfor_rows_n<F>,
				   dim3((static_cast<unsigned>(nx) + tb - 1) / tb, static_cast<unsigned>(bx.length(1)), static_cast<unsigned>(bx.length(2) * ncomp)), dim3(tb), 0,
				   nullptr, bx, f);
	} else {
		hipLaunchKernelGGL(qk_parfor_kernel_n<F>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, bx, ncomp, f);
	}
	qk_check_launch("amrex::ParallelFor");
}
// amrex::ParallelForRNG / amrex::Random: one counter-based stream per cell (splitmix64 of the flat cell index and a draw counter)
struct RandomEngine {
	unsigned long long key = 0, ctr = 0;
};
QK_HD inline auto Random(RandomEngine const &e) -> Real
{
	auto &m = const_cast<RandomEngine &>(e);
	unsigned long long z = m.key + 0x9E3779B97F4A7C15ULL * (++m.ctr);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	z ^= z >> 31;
	return static_cast<Real>(z >> 11) * (1.0 / 9007199254740992.0);
}
// amrex::InitRandom / Random() / RandomPoisson on the host (AMReX_Random.H): one seeded generator per process; every rank seeded alike draws the
// same sequence, which the problems that use it rely on ("all ranks should produce the same values")
inline auto qk_host_rng() -> std::mt19937_64 &
{
	static std::mt19937_64 g(42);
	return g;
}
inline void InitRandom(unsigned long long seed, int /*nprocs*/ = 1) { qk_host_rng().seed(seed); }
inline auto Random() -> Real { return std::uniform_real_distribution<Real>(0.0, 1.0)(qk_host_rng()); }
inline auto RandomPoisson(Real lambda) -> unsigned int { return static_cast<unsigned int>(std::poisson_distribution<long>(lambda)(qk_host_rng())); }
// key of a cell's stream: splitmix64 over (launch number, i, j, k) — every ParallelForRNG call draws a different field (a global launch counter,
// the same on every rank as long as the ranks make the same calls; the cell index is global, so the field does not depend on the box layout),
// negative / ghost indices and indices beyond 2^21 do not alias
QK_HD inline auto qk_mix64(unsigned long long z) -> unsigned long long
{
	z += 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
inline auto qk_rng_launch_counter() -> unsigned long long &
{
	static unsigned long long n = 0;
	return n;
}
template <typename F> void ParallelForRNG(Box const &bx, F const &f)
{
	unsigned long long const launch = ++qk_rng_launch_counter();
	ParallelFor(bx, [=] __device__(int i, int j, int k) {
		RandomEngine e;
		unsigned long long h = qk_mix64(launch);
		h = qk_mix64(h ^ static_cast<unsigned long long>(static_cast<long long>(i)));
		h = qk_mix64(h ^ static_cast<unsigned long long>(static_cast<long long>(j)));
		h = qk_mix64(h ^ static_cast<unsigned long long>(static_cast<long long>(k)));
		e.key = h;
		f(i, j, k, e);
	});
}

namespace BCType
{
enum mathematicalBndryTypes : int { reflect_odd = -1, int_dir = 0, reflect_even = 1, foextrap = 2, ext_dir = 3, hoextrap = 4 };
}

class BCRec
{
      public:
	int bc[6] = {0, 0, 0, 0, 0, 0};
	void setLo(int dir, int type) { bc[dir] = type; }
	void setHi(int dir, int type) { bc[3 + dir] = type; }
	[[nodiscard]] QK_HD auto lo(int dir) const -> int { return bc[dir]; }
	[[nodiscard]] QK_HD auto hi(int dir) const -> int { return bc[3 + dir]; }
};

struct RealBoxData { // amrex::RealBox as GeometryData::prob_domain carries it
	GpuArray<Real, AMREX_SPACEDIM> xlo{}, xhi{};
	[[nodiscard]] QK_HD auto length(int d) const -> Real { return xhi[d] - xlo[d]; }
	[[nodiscard]] QK_HD auto lo(int d) const -> Real { return xlo[d]; }
	[[nodiscard]] QK_HD auto hi(int d) const -> Real { return xhi[d]; }
};
struct GeometryData {
	Box domain;
	GpuArray<Real, AMREX_SPACEDIM> prob_lo{}, prob_hi{}, dx{};
	RealBoxData prob_domain{};
	[[nodiscard]] QK_HD auto Domain() const -> Box const & { return domain; }
	[[nodiscard]] QK_HD auto ProbLo(int d) const -> Real { return prob_lo[d]; }
	[[nodiscard]] QK_HD auto ProbHi(int d) const -> Real { return prob_hi[d]; }
	[[nodiscard]] QK_HD auto ProbLo() const -> Real const * { return prob_lo.data(); }
	[[nodiscard]] QK_HD auto ProbHi() const -> Real const * { return prob_hi.data(); }
	[[nodiscard]] QK_HD auto CellSize(int d) const -> Real { return dx[d]; }
	[[nodiscard]] QK_HD auto CellSize() const -> Real const * { return dx.data(); }
};

class Geometry
{
      public:
	Box domain;
	GpuArray<Real, AMREX_SPACEDIM> prob_lo{}, prob_hi{}, dx{};
	int periodic[3] = {0, 0, 0};
	Geometry() = default;
	explicit Geometry(Box const &dom) : domain(dom) {} // amrex::Geometry(domain): the physical box comes from the deck (geometry.prob_lo / prob_hi)
	[[nodiscard]] auto Domain() const -> Box const & { return domain; }
	[[nodiscard]] auto CellSizeArray() const -> GpuArray<Real, AMREX_SPACEDIM> { return dx; }
	[[nodiscard]] auto ProbLoArray() const -> GpuArray<Real, AMREX_SPACEDIM> { return prob_lo; }
	[[nodiscard]] auto ProbHiArray() const -> GpuArray<Real, AMREX_SPACEDIM> { return prob_hi; }
	[[nodiscard]] auto ProbLo(int d) const -> Real { return prob_lo[d]; }
	[[nodiscard]] auto CellSize(int d) const -> Real { return dx[d]; }
	[[nodiscard]] auto ProbLength(int d) const -> Real { return prob_hi[d] - prob_lo[d]; }
	[[nodiscard]] auto ProbSize() const -> Real
	{
		Real v = 1.0;
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			v *= prob_hi[d] - prob_lo[d];
		}
		return v;
	}
	[[nodiscard]] auto isPeriodic(int d) const -> bool { return periodic[d] != 0; }
	[[nodiscard]] auto isAllPeriodic() const -> bool
	{
		bool a = true;
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			a = a && isPeriodic(d);
		}
		return a;
	}
	[[nodiscard]] auto data() const -> GeometryData { return {domain, prob_lo, prob_hi, dx, RealBoxData{prob_lo, prob_hi}}; }
};

// ParmParse: `key = v1 v2 ...` decks (# comments) + command-line overrides, with prefixes
class ParmParse
{
      public:
	explicit ParmParse(std::string prefix = "") : prefix_(std::move(prefix)) {}
	static auto table() -> std::map<std::string, std::vector<std::string>> &
	{
		static std::map<std::string, std::vector<std::string>> t;
		return t;
	}
	static void addLine(std::string line)
	{
		auto const hash = line.find('#');
		if (hash != std::string::npos) {
			line = line.substr(0, hash);
		}
		auto const eq = line.find('=');
		if (eq == std::string::npos) {
			return;
		}
		std::istringstream k(line.substr(0, eq));
		std::string key;
		k >> key;
		std::istringstream v(line.substr(eq + 1));
		std::vector<std::string> vals;
		for (std::string tok; v >> tok;) {
			// a value in double quotes is the text between them (AMReX's ParmParse: `file = "./table.h5"`); quoted values with blanks inside
			// do not occur in the reference's decks
			if (tok.size() >= 2 && tok.front() == '"' && tok.back() == '"') {
				tok = tok.substr(1, tok.size() - 2);
			}
			vals.push_back(tok);
		}
		if (!key.empty()) {
			table()[key] = vals;
		}
	}
	// amrex::Initialize(argc, argv): argv[1] = deck (optional), remaining `key=value` overrides
	static void Initialize(int argc, char **argv)
	{
		int first = 1;
		if (argc > 1 && std::string(argv[1]).find('=') == std::string::npos) {
			std::ifstream f(argv[1]);
			if (!f.is_open()) {
				Abort(std::string("cannot open input deck ") + argv[1]);
			}
			for (std::string line; std::getline(f, line);) {
				addLine(line);
			}
			first = 2;
		}
		for (int a = first; a < argc; ++a) {
			addLine(argv[a]);
		}
	}
	template <typename T> auto query(std::string const &name, T &val) const -> bool
	{
		auto it = table().find(full(name));
		if (it == table().end() || it->second.empty()) {
			return false;
		}
		if constexpr (std::is_same_v<T, bool>) { // amrex::ParmParse reads a bool from true / t / false / f (any case) or a number
			std::string w = it->second[0];
			for (auto &c : w) {
				c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
			}
			if (w == "true" || w == "t") {
				val = true;
			} else if (w == "false" || w == "f") {
				val = false;
			} else {
				double num = 0.0;
				std::istringstream s(w);
				s >> num;
				val = (num != 0.0);
			}
			return true;
		} else {
			std::istringstream s(it->second[0]);
			s >> val;
			return true;
		}
	}
	template <typename T> auto queryarr(std::string const &name, std::vector<T> &vals) const -> bool
	{
		auto it = table().find(full(name));
		if (it == table().end()) {
			return false;
		}
		vals.clear();
		for (auto const &tok : it->second) {
			std::istringstream s(tok);
			T v;
			s >> v;
			vals.push_back(v);
		}
		return true;
	}
	template <typename T> void add(std::string const &name, T const &val)
	{
		std::ostringstream s;
		s.precision(17);
		s << val;
		table()[full(name)] = {s.str()};
	}

      private:
	std::string prefix_;
	[[nodiscard]] auto full(std::string const &n) const -> std::string { return prefix_.empty() ? n : prefix_ + "." + n; }
};

#define QK_HOST_HIP_DEFINED_BELOW 1
// (every synchronisation, copy and allocation of the mirror goes through here: a pack of batched ParallelFor calls still pending — qk_parfor_batch_scope —
// leaves first, so that a hook which launches, synchronises and reads on the host sees its launches in order)
#define QK_HOST_HIP(expr)                                                                                                                            \
	do {                                                                                                                                         \
		::amrex::qk_parfor_batch_flush();                                                                                                    \
		hipError_t e_ = (expr);                                                                                                              \
		if (e_ != hipSuccess) {                                                                                                              \
			amrex::Abort(std::string(#expr) + ": " + hipGetErrorString(e_));                                                             \
		}                                                                                                                                    \
	} while (0)

// The device arena of the host mirror (what amrex::The_Arena() is to the reference: src/main.cpp initialises AMReX with its pooled arena, and every
// MultiFab of a regrid comes out of it).  hipMalloc / hipFree cost tens to hundreds of microseconds each and hipFree waits for the device; a
// hierarchy that regrids every other step re-creates the tag arrays, the fine-level states and their plans every time.  Blocks are kept in exact-size
// free lists instead (a regrid asks for the sizes the one before released).  A released block may still be read by kernels in flight: it becomes
// reusable when an event recorded on the null stream at release time has completed.  The blocking streams of the host mirror (the compute stream)
// order themselves with the null stream; a NON-blocking stream that touches fab data — the RCCL stream of qk_comm.hpp — is registered
// (alsoWaitFor) and the null stream is made to wait for what it has queued before the event is recorded.  A reused block is not fresh from
// hipMalloc: nothing may rely on zeroed memory (FabArray fills what it needs).  QK_DEVICE_ARENA=0: plain hipMalloc / hipFree.
class DeviceArena
{
      public:
	static auto get() -> DeviceArena &
	{
		static auto *a = new DeviceArena; // (never destroyed: objects with static storage may release blocks after main returns)
		return *a;
	}
	auto alloc(std::size_t bytes) -> void *
	{
		std::size_t const n = roundUp(bytes);
		void *p = nullptr;
		if (pooled_) {
			reclaim(0);
			auto it = free_.find(n);
			if (it == free_.end() || it->second.empty()) {
				reclaim(n); // wait for a released block of this size rather than growing
				it = free_.find(n);
			}
			if (it != free_.end() && !it->second.empty()) {
				p = it->second.back();
				it->second.pop_back();
				cached_ -= n;
				return p;
			}
		}
		if (hipMalloc(&p, n) != hipSuccess) {
			trim(); // give the cache back to the driver and try once more
			hipError_t const e = hipMalloc(&p, n);
			if (e != hipSuccess) {
				std::fprintf(stderr, "DeviceArena: hipMalloc(%zu): %s\n", n, hipGetErrorString(e));
				std::abort();
			}
		}
		if (pooled_) {
			size_[p] = n;
		}
		return p;
	}
	// a non-blocking stream whose work may read or write arena blocks (the null stream's release event does not cover it by itself)
	void alsoWaitFor(hipStream_t s)
	{
		if (s != nullptr && std::find(extra_.begin(), extra_.end(), s) == extra_.end()) {
			extra_.push_back(s);
		}
	}
	void free(void *p)
	{
		if (p == nullptr) {
			return;
		}
		auto const it = size_.find(p);
		if (!pooled_ || it == size_.end()) {
			(void)hipFree(p);
			return;
		}
		alsoWaitFor(qkhost::Comm::get().commStream());
		for (hipStream_t s : extra_) { // the null stream follows whatever the registered streams have queued so far
			hipEvent_t es = nullptr;
			if (hipEventCreateWithFlags(&es, hipEventDisableTiming) == hipSuccess) {
				if (hipEventRecord(es, s) == hipSuccess) {
					(void)hipStreamWaitEvent(nullptr, es, 0);
				}
				(void)hipEventDestroy(es);
			}
		}
		hipEvent_t ev = nullptr;
		if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, nullptr) != hipSuccess) {
			(void)hipGetLastError(); // (the runtime is shutting down: nothing reuses the block)
			return;
		}
		pending_.push_back({p, it->second, ev});
	}

      private:
	struct Pending {
		void *p;
		std::size_t n;
		hipEvent_t ev;
	};
	DeviceArena()
	{
		char const *e = std::getenv("QK_DEVICE_ARENA");
		pooled_ = (e == nullptr) || std::atoi(e) != 0;
	}
	static auto roundUp(std::size_t b) -> std::size_t { return (std::max<std::size_t>(b, 1) + 511) / 512 * 512; }
	// released blocks whose event has completed join the free lists; waitFor != 0: block on the oldest pending release of that size
	void reclaim(std::size_t waitFor)
	{
		for (std::size_t k = 0; k < pending_.size();) {
			Pending const q = pending_[k];
			bool done = hipEventQuery(q.ev) == hipSuccess;
			if (!done && waitFor != 0 && q.n == waitFor) {
				done = hipEventSynchronize(q.ev) == hipSuccess;
				waitFor = 0;
			}
			if (done) {
				(void)hipEventDestroy(q.ev);
				free_[q.n].push_back(q.p);
				cached_ += q.n;
				pending_[k] = pending_.back();
				pending_.pop_back();
			} else {
				++k;
			}
		}
		(void)hipGetLastError(); // (hipEventQuery reports hipErrorNotReady through the sticky error too)
		if (cached_ > (static_cast<std::size_t>(8) << 30)) {
			trim();
		}
	}
	void trim()
	{
		for (auto &kv : free_) {
			for (void *p : kv.second) {
				size_.erase(p);
				(void)hipFree(p);
			}
			kv.second.clear();
		}
		cached_ = 0;
	}
	bool pooled_ = true;
	std::vector<hipStream_t> extra_;
	std::map<std::size_t, std::vector<void *>> free_;
	std::map<void *, std::size_t> size_;
	std::vector<Pending> pending_;
	std::size_t cached_ = 0;
};

// amrex::launch(box, f(Box const &tbx)): the reference uses it on single-cell boxes; one thread gets the whole box
template <typename F> __global__ void qk_launch_kernel(Box bx, F f) { f(bx); }
template <typename F> void launch(Box const &bx, F const &f)
{
	qk_parfor_batch_flush();
	hipLaunchKernelGGL(qk_launch_kernel<F>, dim3(1), dim3(1), 0, nullptr, bx, f);
	qk_check_launch("amrex::launch");
}
// amrex::AsyncArray<T>: a device copy of a host array that lives as long as the object
template <typename T> class AsyncArray
{
      public:
	AsyncArray(T const *h, std::size_t n) : n_(n)
	{
		d_ = static_cast<T *>(DeviceArena::get().alloc(sizeof(T) * (n > 0 ? n : 1)));
		QK_HOST_HIP(hipMemcpy(d_, h, sizeof(T) * n, hipMemcpyHostToDevice));
	}
	AsyncArray(AsyncArray const &) = delete;
	auto operator=(AsyncArray const &) -> AsyncArray & = delete;
	~AsyncArray() { DeviceArena::get().free(d_); }
	[[nodiscard]] auto data() const -> T * { return d_; }
	void copyToHost(T *h, std::size_t n) const { QK_HOST_HIP(hipMemcpy(h, d_, sizeof(T) * n, hipMemcpyDeviceToHost)); }

      private:
	T *d_ = nullptr;
	std::size_t n_ = 0;
};

struct DistributionMapping {
	DistributionMapping() = default;
	template <typename BA> explicit DistributionMapping(BA const & /*ba*/) {}
};

// amrex::Gpu: host / device vectors and copies as the reference's problem files use them for their interpolation tables
namespace Gpu
{
template <typename T> using HostVector = std::vector<T>;
struct HostToDevice {
};
struct DeviceToHost {
};
struct DeviceToDevice {
};
static constexpr HostToDevice hostToDevice{};
static constexpr DeviceToHost deviceToHost{};
static constexpr DeviceToDevice deviceToDevice{};
template <typename T> class DeviceVector
{
      public:
	DeviceVector() = default;
	explicit DeviceVector(size_t n) { resize(n); }
	DeviceVector(DeviceVector const &) = delete;
	auto operator=(DeviceVector const &) -> DeviceVector & = delete;
	~DeviceVector()
	{
		if (p_ != nullptr) {
			(void)hipFree(p_);
		}
	}
	void resize(size_t n)
	{
		if (n > cap_) {
			T *q = nullptr;
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&q), sizeof(T) * n));
			if (p_ != nullptr) {
				QK_HOST_HIP(hipMemcpy(q, p_, sizeof(T) * n_, hipMemcpyDeviceToDevice));
				(void)hipFree(p_);
			}
			p_ = q;
			cap_ = n;
		}
		n_ = n;
	}
	[[nodiscard]] auto size() const -> size_t { return n_; }
	[[nodiscard]] auto data() -> T * { return p_; }
	[[nodiscard]] auto data() const -> T const * { return p_; }
	[[nodiscard]] auto dataPtr() -> T * { return p_; }
	[[nodiscard]] auto dataPtr() const -> T const * { return p_; }
	[[nodiscard]] auto begin() -> T * { return p_; }
	[[nodiscard]] auto end() -> T * { return p_ + n_; }
	[[nodiscard]] auto begin() const -> T const * { return p_; }
	[[nodiscard]] auto end() const -> T const * { return p_ + n_; }

      private:
	T *p_ = nullptr;
	size_t n_ = 0, cap_ = 0;
};
template <typename It, typename Out> void copy(HostToDevice /*tag*/, It first, It last, Out out)
{
	auto const n = static_cast<size_t>(last - first);
	if (n > 0) {
		QK_HOST_HIP(hipMemcpy(&*out, &*first, sizeof(*first) * n, hipMemcpyHostToDevice));
	}
}
template <typename It, typename Out> void copy(DeviceToHost /*tag*/, It first, It last, Out out)
{
	auto const n = static_cast<size_t>(last - first);
	if (n > 0) {
		QK_HOST_HIP(hipMemcpy(&*out, &*first, sizeof(*first) * n, hipMemcpyDeviceToHost));
	}
}
template <typename Tag, typename It, typename Out> void copyAsync(Tag tag, It first, It last, Out out) { copy(tag, first, last, out); }
inline void htod_memcpy(void *dst, void const *src, size_t bytes) { QK_HOST_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); }
inline void dtoh_memcpy(void *dst, void const *src, size_t bytes) { QK_HOST_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); }
inline void htod_memcpy_async(void *dst, void const *src, size_t bytes) { htod_memcpy(dst, src, bytes); }
inline void streamSynchronize() { QK_HOST_HIP(hipDeviceSynchronize()); }
inline void streamSynchronizeAll() { QK_HOST_HIP(hipDeviceSynchronize()); }
inline void synchronize() { QK_HOST_HIP(hipDeviceSynchronize()); }
struct Device {
	static void streamSynchronize() { QK_HOST_HIP(hipDeviceSynchronize()); }
	static void synchronize() { QK_HOST_HIP(hipDeviceSynchronize()); }
};
} // namespace Gpu

// amrex::TableData<T, N> (AMReX_TableData.H): an N-dimensional table with inclusive index bounds, first index fastest, in device memory or — built
// with The_Pinned_Arena() — in host memory; table() / const_table() hand out the accessor a kernel captures by value.  Only what the reference's
// problem files use: N = 1 ... 3, copy() from a host table, the accessors.
struct Arena {
	bool host;
};
inline auto The_Pinned_Arena() -> Arena *
{
	static Arena a{true};
	return &a;
}
inline auto The_Arena() -> Arena *
{
	static Arena a{false};
	return &a;
}
template <typename T> struct Table1D {
	T *p = nullptr;
	int begin = 0, end = 0; // [begin, end)
	QK_HD auto operator()(int i) const -> T & { return p[i - begin]; }
};
template <typename T> struct Table2D {
	T *p = nullptr;
	Long jstride = 0;
	int lo0 = 0, lo1 = 0;
	QK_HD auto operator()(int i, int j) const -> T & { return p[(i - lo0) + jstride * (j - lo1)]; }
};
template <typename T> struct Table3D {
	T *p = nullptr;
	Long jstride = 0, kstride = 0;
	int lo0 = 0, lo1 = 0, lo2 = 0;
	GpuArray<int, 3> begin{{0, 0, 0}}, end{{0, 0, 0}}; // [begin, end) as AMReX names them
	Table3D() = default;
	// amrex::Table3D(p, lo, hi): hi is one past the last index
	Table3D(T *ptr, GpuArray<int, 3> const &lo, GpuArray<int, 3> const &hi)
	    : p(ptr), jstride(hi[0] - lo[0]), kstride(static_cast<Long>(hi[0] - lo[0]) * (hi[1] - lo[1])), lo0(lo[0]), lo1(lo[1]), lo2(lo[2]), begin(lo), end(hi)
	{
	}
	QK_HD auto operator()(int i, int j, int k) const -> T & { return p[(i - lo0) + jstride * (j - lo1) + kstride * (k - lo2)]; }
};
template <typename T, int N> struct TableAccessor;
template <typename T> struct TableAccessor<T, 1> {
	using type = Table1D<T>;
};
template <typename T> struct TableAccessor<T, 2> {
	using type = Table2D<T>;
};
template <typename T> struct TableAccessor<T, 3> {
	using type = Table3D<T>;
};
template <typename T, int N> class TableData
{
	static_assert(N >= 1 && N <= 3, "amrex_mini: TableData is built for one, two or three indices");

      public:
	TableData() = default;
	TableData(TableData &&o) noexcept : lo_(o.lo_), hi_(o.hi_), host_(o.host_), n_(o.n_), d_(o.d_) { o.d_ = nullptr; }
	[[nodiscard]] auto lo() const -> Array<int, N> const & { return lo_; }
	[[nodiscard]] auto hi() const -> Array<int, N> const & { return hi_; }
	void resize(Array<int, N> const &lo, Array<int, N> const &hi, Arena *arena = nullptr)
	{
		if (d_ != nullptr) {
			(void)(host_ ? hipHostFree(d_) : hipFree(d_));
			d_ = nullptr;
		}
		new (this) TableData(lo, hi, arena);
	}
	TableData(Array<int, N> const &lo, Array<int, N> const &hi, Arena *arena = nullptr) : lo_(lo), hi_(hi), host_(arena != nullptr && arena->host)
	{
		n_ = 1;
		for (int d = 0; d < N; ++d) {
			n_ *= static_cast<Long>(hi[d] - lo[d] + 1);
		}
		size_t const bytes = sizeof(T) * static_cast<size_t>(std::max<Long>(n_, 1));
		if (host_) { // The_Pinned_Arena(): page-locked host memory, the same pointer on the host and in kernels
			QK_HOST_HIP(hipHostMalloc(reinterpret_cast<void **>(&d_), bytes, hipHostMallocDefault));
			std::memset(d_, 0, bytes);
		} else {
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_), bytes));
		}
	}
	TableData(TableData const &) = delete;
	auto operator=(TableData const &) -> TableData & = delete;
	~TableData()
	{
		if (d_ != nullptr) {
			(void)(host_ ? hipHostFree(d_) : hipFree(d_));
		}
	}
	[[nodiscard]] auto table() -> typename TableAccessor<T, N>::type { return make<T>(); }
	[[nodiscard]] auto table() const -> typename TableAccessor<T const, N>::type { return make<T const>(); }
	[[nodiscard]] auto const_table() const -> typename TableAccessor<T const, N>::type { return make<T const>(); }
	[[nodiscard]] auto size() const -> Long { return n_; }
	void copy(TableData const &src) // same bounds; any combination of host and device storage
	{
		AMREX_ALWAYS_ASSERT(src.n_ == n_);
		QK_HOST_HIP(hipMemcpy(data(), src.data(), sizeof(T) * static_cast<size_t>(n_), hipMemcpyDefault));
	}

      private:
	[[nodiscard]] auto data() const -> T * { return d_; }
	template <typename U> [[nodiscard]] auto make() const -> typename TableAccessor<U, N>::type
	{
		typename TableAccessor<U, N>::type t;
		t.p = data();
		if constexpr (N == 1) {
			t.begin = lo_[0];
			t.end = hi_[0] + 1;
		} else {
			t.jstride = hi_[0] - lo_[0] + 1;
			t.lo0 = lo_[0];
			t.lo1 = lo_[1];
			if constexpr (N == 3) {
				t.kstride = t.jstride * (hi_[1] - lo_[1] + 1);
				t.lo2 = lo_[2];
				t.begin = {{lo_[0], lo_[1], lo_[2]}};
				t.end = {{hi_[0] + 1, hi_[1] + 1, hi_[2] + 1}};
			}
		}
		return t;
	}
	Array<int, N> lo_{}, hi_{};
	bool host_ = false;
	Long n_ = 0;
	T *d_ = nullptr;
};

// one process per GPU; the problem-side reductions of the reference are over one rank here
// one process per GPU: the reductions a problem file or the driver performs over ranks go through qkhost::Comm (qk_comm.hpp); with one rank
// every function returns its argument
namespace ParallelDescriptor
{
inline auto IOProcessor() -> bool { return qkhost::Comm::get().rank == 0; }
inline auto IOProcessorNumber() -> int { return 0; }
inline auto MyProc() -> int { return qkhost::Comm::get().rank; }
inline auto NProcs() -> int { return qkhost::Comm::get().size; }
inline void Barrier() { qkhost::Comm::get().barrier(); }
template <typename T> void ReduceRealSum(T &v) { v = static_cast<T>(qkhost::Comm::get().allReduceSum(static_cast<double>(v))); }
template <typename T> void ReduceRealMax(T &v) { v = static_cast<T>(qkhost::Comm::get().allReduceMax(static_cast<double>(v))); }
template <typename T> void ReduceRealMin(T &v) { v = static_cast<T>(qkhost::Comm::get().allReduceMin(static_cast<double>(v))); }
inline void ReduceIntSum(int &v) { v = static_cast<int>(qkhost::Comm::get().allReduceSum(static_cast<int64_t>(v))); }
inline void ReduceLongSum(long &v) { v = static_cast<long>(qkhost::Comm::get().allReduceSum(static_cast<int64_t>(v))); }
inline void Abort() { std::exit(2); }
// MPI plumbing a problem may name: the type map of a broadcast, the I/O rank, the communicator (AMReX_ParallelDescriptor.H)
template <typename T> struct Mpi_typemap {
	static auto type() -> int { return static_cast<int>(sizeof(T)); }
};
constexpr int ioProcessor = 0;
inline auto Communicator() -> int { return 0; }
template <typename T> void Bcast(T *v, size_t n, int /*type*/, int root, int /*comm*/)
{
	for (size_t i = 0; i < n; ++i) { // (a sum of the root's value and zeros: the collectives this host has)
		double const mine = (qkhost::Comm::get().rank == root && v[i] == v[i]) ? static_cast<double>(v[i]) : 0.0;
		double const isnan_root = (qkhost::Comm::get().rank == root && v[i] != v[i]) ? 1.0 : 0.0;
		double const s = qkhost::Comm::get().allReduceSum(mine);
		v[i] = (qkhost::Comm::get().allReduceSum(isnan_root) > 0) ? static_cast<T>(std::numeric_limits<double>::quiet_NaN()) : static_cast<T>(s);
	}
}
} // namespace ParallelDescriptor
namespace ParallelContext
{
inline auto CommunicatorSub() -> int { return 0; }
inline auto IOProcessorSub() -> bool { return qkhost::Comm::get().rank == 0; }
} // namespace ParallelContext
namespace ParallelAllReduce
{
template <typename T> void Sum(T &v, int /*comm*/) { ParallelDescriptor::ReduceRealSum(v); }
template <typename T> void Max(T &v, int /*comm*/) { ParallelDescriptor::ReduceRealMax(v); }
template <typename T> void Min(T &v, int /*comm*/) { ParallelDescriptor::ReduceRealMin(v); }
} // namespace ParallelAllReduce
template <typename F> void LoopOnCpu(Box const &bx, F &&f) { HostFor(bx, f); }
// device fill by a kernel on the default stream
template <typename T> __global__ void qk_fill_kernel(T *p, Long n, T v)
{
	for (Long t = static_cast<Long>(blockIdx.x) * blockDim.x + threadIdx.x; t < n; t += static_cast<Long>(gridDim.x) * blockDim.x) {
		p[t] = v;
	}
}
template <typename T> void qk_device_fill(T *p, Long n, T v)
{
	if (n <= 0) {
		return;
	}
	unsigned const blocks = static_cast<unsigned>(std::min<Long>((n + 255) / 256, 65535));
	qk_parfor_batch_flush();
	hipLaunchKernelGGL(qk_fill_kernel<T>, dim3(blocks), dim3(256), 0, nullptr, p, n, v);
	qk_check_launch("qk_device_fill");
}
class BoxArray : public std::vector<Box>
{
      public:
	using std::vector<Box>::vector;
	BoxArray() = default;
	BoxArray(std::vector<Box> const &v) : std::vector<Box>(v) {} // NOLINT
	explicit BoxArray(Box const &b) : std::vector<Box>{b} {}
	void maxSize(int /*n*/) {}
	int facedir = -1; // index type of the boxes: cell-centred, or nodal in this direction (the BoxArray of a face-centred MultiFab)
};

// FabArray<T> with device storage: one allocation for all boxes + the device table of Array4 descriptors
template <typename T> class FabArrayT
{
      public:
	FabArrayT() = default;
	FabArrayT(std::vector<Box> const &ba, int ncomp, int nghost, int facedir = -1) { define(ba, ncomp, nghost, facedir); }
	FabArrayT(std::vector<Box> const &ba, DistributionMapping const & /*dm*/, int ncomp, int nghost) { define(ba, ncomp, nghost, -1); }
	// (`ba` as boxArray() of another array hands it out: index type included)
	FabArrayT(BoxArray const &ba, DistributionMapping const & /*dm*/, int ncomp, int nghost) { define(ba, ncomp, nghost, ba.facedir); }
	FabArrayT(FabArrayT const &) = delete;
	auto operator=(FabArrayT const &) -> FabArrayT & = delete;
	FabArrayT(FabArrayT &&o) noexcept { *this = std::move(o); }
	auto operator=(FabArrayT &&o) noexcept -> FabArrayT &
	{
		release();
		boxes_ = std::move(o.boxes_);
		fabboxes_ = std::move(o.fabboxes_);
		offsets_ = std::move(o.offsets_);
		ncomp_ = o.ncomp_;
		nghost_ = o.nghost_;
		facedir_ = o.facedir_;
		total_ = o.total_;
		d_data_ = o.d_data_;
		d_table_ = o.d_table_;
		o.d_data_ = nullptr;
		o.d_table_ = nullptr;
		return *this;
	}
	~FabArrayT() { release(); }

	void define(std::vector<Box> const &ba, int ncomp, int nghost, int facedir = -1)
	{
		release();
		static_cast<std::vector<Box> &>(boxes_) = ba;
		boxes_.facedir = facedir;
		ncomp_ = ncomp;
		nghost_ = nghost;
		facedir_ = facedir;
		fabboxes_.clear();
		offsets_.clear();
		total_ = 0;
		std::vector<Array4<T>> tab;
		for (auto const &b : ba) {
			Box fb = b;
			if (facedir >= 0) {
				fb.hi[facedir] += 1;
			}
			fb = grow(fb, nghost);
			fabboxes_.push_back(fb);
			offsets_.push_back(total_);
			total_ += fb.numPts() * ncomp;
		}
		// Some of the reference's problem files initialise the radiation components of a state that has none (HydroBlast2D's
		// setInitialConditionsOnGrid writes radEnergy_index .. x3RadFlux_index with is_radiation_enabled = false): four components past the end
		// of the fab.  AMReX's arena hands out fabs from large chunks, so the stray writes land in memory nobody reads; here the last fab of a
		// cell-centred state array is followed by the same amount of slack instead of the end of the allocation.
		Long slack = 0;
		if (facedir < 0 && ncomp >= 5) {
			for (auto const &fb : fabboxes_) {
				slack = std::max<Long>(slack, 4 * fb.numPts());
			}
		}
		d_data_ = static_cast<T *>(DeviceArena::get().alloc(sizeof(T) * static_cast<std::size_t>(std::max<Long>(total_ + slack, 1))));
		// fresh storage reads as zero (what the Python drivers' MultiFab(fill = 0) gives); QK_POISON=1 fills it with NaN bit patterns
		// instead, which makes any read of a cell that was never written visible in the results (debugging aid)
		// fresh storage reads as zero (what the Python drivers' MultiFab(fill = 0) gives); QK_POISON=1 fills it with NaN bit patterns instead, which
		// makes any read of a cell that was never written visible in the results (debugging aid).  Fills are kernels on the default stream, like
		// everything else that touches the array.
		if (std::getenv("QK_POISON") != nullptr) {
			qk_device_fill(reinterpret_cast<unsigned char *>(d_data_), static_cast<Long>(sizeof(T)) * std::max<Long>(total_, 1), static_cast<unsigned char>(0xFF));
		} else {
			qk_device_fill(d_data_, std::max<Long>(total_, 1), T{});
		}
		for (size_t n = 0; n < ba.size(); ++n) {
			tab.emplace_back(d_data_ + offsets_[n], fabboxes_[n], ncomp);
		}
		d_table_ = static_cast<Array4<T> *>(DeviceArena::get().alloc(sizeof(Array4<T>) * std::max<size_t>(ba.size(), 1))); // (never null: a level may be empty on this rank)
		QK_HOST_HIP(hipMemcpy(d_table_, tab.data(), sizeof(Array4<T>) * ba.size(), hipMemcpyHostToDevice));
	}
	[[nodiscard]] auto size() const -> int { return static_cast<int>(boxes_.size()); }
	[[nodiscard]] auto faceDir() const -> int { return facedir_; }
	[[nodiscard]] auto nComp() const -> int { return ncomp_; }
	[[nodiscard]] auto nGrow() const -> int { return nghost_; }
	[[nodiscard]] auto nGrowVect() const -> IntVect
	{
		return {nghost_, AMREX_SPACEDIM >= 2 ? nghost_ : 0, AMREX_SPACEDIM >= 3 ? nghost_ : 0};
	}
	[[nodiscard]] auto boxArray() const -> BoxArray const & { return boxes_; }
	[[nodiscard]] auto validbox(int b) const -> Box const & { return boxes_[b]; }
	[[nodiscard]] auto fabbox(int b) const -> Box const & { return fabboxes_[b]; }
	// host copy of one descriptor (device data pointer)
	[[nodiscard]] auto array(int b) const -> Array4<T> { return Array4<T>(d_data_ + offsets_[b], fabboxes_[b], ncomp_); }
	[[nodiscard]] auto const_array(int b) const -> Array4<T const> { return Array4<T const>(d_data_ + offsets_[b], fabboxes_[b], ncomp_); }
	template <typename It, typename = decltype(std::declval<It>().index())> [[nodiscard]] auto array(It const &mfi) const -> Array4<T> { return array(mfi.index()); }
	template <typename It, typename = decltype(std::declval<It>().index())> [[nodiscard]] auto const_array(It const &mfi) const -> Array4<T const>
	{
		return const_array(mfi.index());
	}
	[[nodiscard]] auto DistributionMap() const -> DistributionMapping { return DistributionMapping{}; }
	// device pointer to the descriptor table (MultiFab::arrays())
	[[nodiscard]] auto arrays() const -> Array4<T> * { return d_table_; }
	// Restrict the DEVICE descriptors (arrays()) to a window of every fab: the same memory and strides, begin / end cropped to win[b] (an empty box:
	// no cell).  For byte arrays a kernel only consults inside a bounding box (qk_hydro_stage_args::flux_mask); array(b) keeps describing the whole fab.
	void cropDeviceTable(std::vector<Box> const &win)
	{
		std::vector<Array4<T>> tab;
		for (int b = 0; b < size(); ++b) {
			Array4<T> a = array(b);
			Box const &w = win[static_cast<size_t>(b)];
			if (w.ok()) {
				a.p += (w.lo[0] - a.begin.x) + a.jstride * (w.lo[1] - a.begin.y) + a.kstride * (w.lo[2] - a.begin.z);
				a.begin = Dim3{w.lo[0], w.lo[1], w.lo[2]};
				a.end = Dim3{w.hi[0] + 1, w.hi[1] + 1, w.hi[2] + 1};
			} else {
				a.end = a.begin;
			}
			tab.push_back(a);
		}
		if (!tab.empty()) {
			QK_HOST_HIP(hipMemcpy(d_table_, tab.data(), sizeof(Array4<T>) * tab.size(), hipMemcpyHostToDevice));
		}
	}
	[[nodiscard]] auto const_arrays() const -> Array4<T const> const * { return reinterpret_cast<Array4<T const> const *>(d_table_); }
	void setVal(T v) { qk_device_fill(d_data_, total_, v); }
	void setZeroAsync(hipStream_t /*s*/) { qk_device_fill(d_data_, total_, T{}); }
	// host staging copies of one fab
	[[nodiscard]] auto copyToHost(int b) const -> std::vector<T>
	{
		std::vector<T> h(static_cast<size_t>(fabboxes_[b].numPts()) * ncomp_);
		QK_HOST_HIP(hipMemcpy(h.data(), d_data_ + offsets_[b], sizeof(T) * h.size(), hipMemcpyDeviceToHost));
		return h;
	}
	void copyFromHost(int b, std::vector<T> const &h) { QK_HOST_HIP(hipMemcpy(d_data_ + offsets_[b], h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice)); }
	// FabArray::ParallelCopy(src): valid cells of `src` into the cells of this array they cover (all components; boxes of this rank)
	void ParallelCopy(FabArrayT const &src);
	static void Copy(FabArrayT &dst, FabArrayT const &src)
	{
		QK_HOST_HIP(hipMemcpy(dst.d_data_, src.d_data_, sizeof(T) * src.total_, hipMemcpyDeviceToDevice));
	}
	// MultiFab::Copy(dst, src, srccomp, dstcomp, numcomp, nghost).  Components are outermost, so the copy is one contiguous
	// block per box; it always carries the ghost cells along (AMReX copies valid + nghost cells) — every caller refills the
	// ghost cells of `dst` before reading them.
	static void Copy(FabArrayT &dst, FabArrayT const &src, int srccomp, int dstcomp, int numcomp, int /*nghost*/)
	{
		for (int b = 0; b < src.size(); ++b) {
			size_t const n = static_cast<size_t>(src.fabboxes_[b].numPts());
			QK_HOST_HIP(hipMemcpy(dst.d_data_ + dst.offsets_[b] + n * dstcomp, src.d_data_ + src.offsets_[b] + n * srccomp, sizeof(T) * n * numcomp,
					      hipMemcpyDeviceToDevice));
		}
	}
	// MultiFab::Subtract(dst, src, srccomp, dstcomp, numcomp, nghost): dst -= src (whole fabs: same layout required)
	static void Subtract(FabArrayT &dst, FabArrayT const &src, int srccomp, int dstcomp, int numcomp, int /*nghost*/)
	{
		for (int b = 0; b < src.size(); ++b) {
			auto const d = dst.array(b);
			auto const s = src.const_array(b);
			ParallelFor(dst.fabbox(b), numcomp, [=] __device__(int i, int j, int k, int n) { d(i, j, k, dstcomp + n) -= s(i, j, k, srccomp + n); });
		}
	}
	// MultiFab::Multiply(dst, src, srccomp, dstcomp, numcomp, nghost): dst *= src (whole fabs: same layout required)
	static void Multiply(FabArrayT &dst, FabArrayT const &src, int srccomp, int dstcomp, int numcomp, int /*nghost*/)
	{
		for (int b = 0; b < src.size(); ++b) {
			auto const d = dst.array(b);
			auto const s = src.const_array(b);
			ParallelFor(dst.fabbox(b), numcomp, [=] __device__(int i, int j, int k, int n) { d(i, j, k, dstcomp + n) *= s(i, j, k, srccomp + n); });
		}
	}
	// MultiFab::norm1(comp): sum of |value| over the valid region (faces of a face-centred array: the nodal valid box), all ranks
	[[nodiscard]] auto norm1(int n) const -> double
	{
		double s = 0;
		QK_HOST_HIP(hipDeviceSynchronize());
		for (int b = 0; b < size(); ++b) {
			auto h = copyToHost(b);
			Array4<T> a(h.data(), fabboxes_[b], ncomp_);
			Box vb = boxes_[b];
			if (facedir_ >= 0) {
				vb.hi[facedir_] += 1;
			}
			HostFor(vb, [&](int i, int j, int k) { s += std::abs(static_cast<double>(a(i, j, k, n))); });
		}
		return qkhost::Comm::get().allReduceSum(s);
	}
	// sum / norm over valid cells of component n (host reduction: diagnostics only, not on the timed path)
	[[nodiscard]] auto sum(int n) const -> double
	{
		// Kahan-compensated: AMReX reduces with a tree on the device, whose rounding error is far below a serial sum's
		double s = 0, comp = 0;
		for (int b = 0; b < size(); ++b) {
			auto h = copyToHost(b);
			Array4<T> a(h.data(), fabboxes_[b], ncomp_);
			HostFor(boxes_[b], [&](int i, int j, int k) {
				double const y = static_cast<double>(a(i, j, k, n)) - comp;
				double const t = s + y;
				comp = (t - s) - y;
				s = t;
			});
		}
		return qkhost::Comm::get().allReduceSum(s); // (MultiFab::sum reduces over all ranks)
	}

      private:
	void release()
	{
		DeviceArena::get().free(d_data_);
		DeviceArena::get().free(d_table_);
		d_data_ = nullptr;
		d_table_ = nullptr;
	}
	BoxArray boxes_;
	std::vector<Box> fabboxes_;
	std::vector<Long> offsets_;
	int ncomp_ = 0, nghost_ = 0, facedir_ = -1;
	Long total_ = 0;
	T *d_data_ = nullptr;
	Array4<T> *d_table_ = nullptr;
};
// amrex::BaseFab<T>: one box of values in host memory (what computePlaneProjection returns: reference src/simulation.hpp:2394-2450)
template <typename T> class BaseFab
{
      public:
	BaseFab() = default;
	BaseFab(Box const &b, int ncomp, Arena * /*arena*/ = nullptr) : box_(b), ncomp_(ncomp), v_(static_cast<size_t>(b.numPts()) * static_cast<size_t>(ncomp), T{}) {}
	[[nodiscard]] auto box() const -> Box const & { return box_; }
	[[nodiscard]] auto nComp() const -> int { return ncomp_; }
	[[nodiscard]] auto size() const -> Long { return static_cast<Long>(v_.size()); }
	[[nodiscard]] auto dataPtr() -> T * { return v_.data(); }
	[[nodiscard]] auto dataPtr() const -> T const * { return v_.data(); }
	[[nodiscard]] auto array() -> Array4<T> { return Array4<T>(v_.data(), box_, ncomp_); }
	[[nodiscard]] auto const_array() const -> Array4<T const> { return Array4<T const>(v_.data(), box_, ncomp_); }

      private:
	Box box_;
	int ncomp_ = 0;
	std::vector<T> v_;
};
// amrex::MFIter over the local boxes of a FabArray
class MFIter
{
      public:
	template <typename FA> explicit MFIter(FA const &fa, bool /*tiling*/ = false) : n_(fa.size()), boxes_(&fa.boxArray()), ng_(fa.nGrow()) {}
	[[nodiscard]] auto isValid() const -> bool { return i_ < n_; }
	void operator++() { ++i_; }
	[[nodiscard]] auto index() const -> int { return i_; }
	[[nodiscard]] auto validbox() const -> Box const & { return (*boxes_)[i_]; }
	[[nodiscard]] auto tilebox() const -> Box const & { return (*boxes_)[i_]; }
	[[nodiscard]] auto fabbox() const -> Box { return grow((*boxes_)[i_], ng_); }

      private:
	int i_ = 0, n_ = 0;
	std::vector<Box> const *boxes_ = nullptr;
	int ng_ = 0;
};
// amrex::ParallelFor(mf, f(box_no, i, j, k)) / (mf, nghost, f): every valid (grown) cell of every local box
template <typename T, typename F> void ParallelFor(FabArrayT<T> const &mf, IntVect const &ng, F const &f)
{
	for (int b = 0; b < mf.size(); ++b) {
		Box bx = mf.validbox(b);
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			bx.lo[d] -= ng[d];
			bx.hi[d] += ng[d];
		}
		ParallelFor(bx, [=] __device__(int i, int j, int k) { f(b, i, j, k); });
	}
}
template <typename T, typename F> void ParallelFor(FabArrayT<T> const &mf, F const &f) { ParallelFor(mf, IntVect(0, 0, 0), f); }
template <typename T> void FabArrayT<T>::ParallelCopy(FabArrayT<T> const &src)
{
	if (qkhost::Comm::get().size > 1) {
		Abort("FabArray::ParallelCopy: not carried across ranks by the host mirror");
	}
	const int nc = ncomp_ < src.ncomp_ ? ncomp_ : src.ncomp_;
	for (int d = 0; d < size(); ++d) {
		for (int s = 0; s < src.size(); ++s) {
			Box const is = fabboxes_[d] & src.boxes_[s];
			if (!is.ok()) {
				continue;
			}
			auto const to = array(d);
			auto const from = src.const_array(s);
			ParallelFor(is, nc, [=] __device__(int i, int j, int k, int n) { to(i, j, k, n) = from(i, j, k, n); });
		}
	}
	// amrex::FabArray::ParallelCopy returns with the data in place: the callers read the destination on the host straight away
	// (HydroRichtmeyerMeshkov's symmetry check does)
	QK_HOST_HIP(hipDeviceSynchronize());
}
using MultiFab = FabArrayT<Real>;
// amrex::ReduceOpSum / ReduceOpMin as tags of computePlaneProjection<ReduceOp>
struct ReduceOpSum {
};
struct ReduceOpMin {
};
// amrex::GetVecOfConstPtrs / amrex::volumeWeightedSum (AMReX_MultiFabUtil.H) for what a simulation object of this mirror holds: ONE level, so no
// finer level masks any of it — the sum over valid cells times the cell volume
template <typename A> auto GetVecOfConstPtrs(A const &a) -> std::vector<FabArrayT<Real> const *>
{
	std::vector<FabArrayT<Real> const *> v;
	for (int i = 0; i < static_cast<int>(a.size()); ++i) {
		v.push_back(&a[i]);
	}
	return v;
}
template <typename G, typename R> auto volumeWeightedSum(std::vector<FabArrayT<Real> const *> const &mf, int comp, G const &geom, R const & /*ratio*/) -> Real
{
	auto const dx = geom[0].CellSizeArray();
	Real vol = 1.0;
	for (int d = 0; d < AMREX_SPACEDIM; ++d) {
		vol *= dx[d];
	}
	return mf[0]->sum(comp) * vol;
}
using iMultiFab = FabArrayT<int>;
// amrex::TagBoxArray: one char per cell (amrex::TagBox::CLEAR = 0, BUF = 1, SET = 2)
using TagBoxArray = FabArrayT<char>;
struct TagBox {
	enum TagVal : char { CLEAR = 0, BUF = 1, SET = 2 };
};

} // namespace amrex

// amrex::literals (AMReX_REAL.H): 0._rt is an amrex::Real.  AMReX declares them so that they are found without a using-directive (the problem files
// write `0._rt` after `using amrex::Real;` only): global scope here.
constexpr auto operator""_rt(long double x) -> amrex::Real { return static_cast<amrex::Real>(x); }
constexpr auto operator""_rt(unsigned long long int x) -> amrex::Real { return static_cast<amrex::Real>(x); }
namespace amrex
{
namespace literals
{
using ::operator""_rt;
}
} // namespace amrex

#endif // QK_HOST_AMREX_MINI_HPP_
