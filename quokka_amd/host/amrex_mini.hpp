// amrex_mini.hpp — the slice of the AMReX API that the reference's hot-path callers and the three config problems use
// (Box, IntVect, Array4, GpuArray, BCRec, Geometry, MultiFab / iMultiFab with device storage, ParmParse, ParallelFor
// on host staging buffers, Print / Abort).  AMReX is an un-vendored submodule of the reference and is absent from this
// image, so problem generators written against the reference's surface are compiled against this header instead.
// Device data live in HIP allocations; the descriptor table of a MultiFab (`arrays()`) is what the C-ABI consumes
// (qk_array4 == amrex::Array4<Real>).  Host C++17, no kernels here.
#ifndef QK_HOST_AMREX_MINI_HPP_
#define QK_HOST_AMREX_MINI_HPP_

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "quokka_amd.h"

#ifndef AMREX_SPACEDIM
#define AMREX_SPACEDIM 3
#endif
#define AMREX_GPU_DEVICE
#define AMREX_GPU_HOST_DEVICE
#define AMREX_FORCE_INLINE inline
#define AMREX_ASSERT(x) ((void)0)
#define AMREX_ALWAYS_ASSERT(x)                                                                                                                       \
	do {                                                                                                                                         \
		if (!(x)) {                                                                                                                          \
			amrex::Abort("assertion failed: " #x);                                                                                       \
		}                                                                                                                                    \
	} while (0)
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

namespace amrex
{
using Real = double;
using Long = long;
template <typename T> using Vector = std::vector<T>;

[[noreturn]] inline void Abort(std::string const &msg)
{
	std::fprintf(stderr, "amrex::Abort: %s\n", msg.c_str());
	std::exit(2);
}
inline void ignore_unused(...) {}

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

struct IntVect {
	int v[3] = {0, 0, 0};
	IntVect() = default;
	IntVect(int i, int j = 0, int k = 0) : v{i, j, k} {}
	auto operator[](int d) const -> int { return v[d]; }
	auto operator[](int d) -> int & { return v[d]; }
	[[nodiscard]] auto toArray() const -> std::array<int, AMREX_SPACEDIM>
	{
		std::array<int, AMREX_SPACEDIM> a{};
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			a[d] = v[d];
		}
		return a;
	}
};

template <typename T, int N> struct GpuArray {
	T arr[N > 0 ? N : 1];
	auto operator[](int i) const -> T const & { return arr[i]; }
	auto operator[](int i) -> T & { return arr[i]; }
	[[nodiscard]] auto begin() const -> T const * { return arr; }
	[[nodiscard]] auto data() const -> T const * { return arr; }
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
	[[nodiscard]] auto smallEnd(int d) const -> int { return lo[d]; }
	[[nodiscard]] auto bigEnd(int d) const -> int { return hi[d]; }
	[[nodiscard]] auto length(int d) const -> int { return hi[d] - lo[d] + 1; }
	[[nodiscard]] auto numPts() const -> Long { return static_cast<Long>(length(0)) * length(1) * length(2); }
	[[nodiscard]] auto loVect3d() const -> GpuArray<int, 3> { return {{lo[0], lo[1], lo[2]}}; }
	[[nodiscard]] auto hiVect3d() const -> GpuArray<int, 3> { return {{hi[0], hi[1], hi[2]}}; }
	[[nodiscard]] auto contains(int i, int j, int k) const -> bool
	{
		return i >= lo[0] && i <= hi[0] && j >= lo[1] && j <= hi[1] && k >= lo[2] && k <= hi[2];
	}
};
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
	Array4(T *ptr, Box const &bx, int nc) : p(ptr), begin{bx.lo[0], bx.lo[1], bx.lo[2]}, end{bx.hi[0] + 1, bx.hi[1] + 1, bx.hi[2] + 1}, ncomp(nc)
	{
		jstride = bx.length(0);
		kstride = jstride * bx.length(1);
		nstride = kstride * bx.length(2);
	}
	auto operator()(int i, int j, int k, int n = 0) const -> T & { return p[(i - begin.x) + jstride * (j - begin.y) + kstride * (k - begin.z) + nstride * n]; }
	[[nodiscard]] auto nComp() const -> int { return ncomp; }
	[[nodiscard]] auto contains(int i, int j, int k) const -> bool
	{
		return i >= begin.x && i < end.x && j >= begin.y && j < end.y && k >= begin.z && k < end.z;
	}
};
static_assert(sizeof(Array4<Real>) == sizeof(qk_array4), "amrex::Array4<Real> must match qk_array4");

// host-side loops standing in for the device ParallelFor on staging buffers
template <typename F> void ParallelFor(Box const &bx, F &&f)
{
	for (int k = bx.lo[2]; k <= bx.hi[2]; ++k) {
		for (int j = bx.lo[1]; j <= bx.hi[1]; ++j) {
			for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) {
				f(i, j, k);
			}
		}
	}
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
	[[nodiscard]] auto lo(int dir) const -> int { return bc[dir]; }
	[[nodiscard]] auto hi(int dir) const -> int { return bc[3 + dir]; }
};

struct GeometryData {
	Box domain;
	GpuArray<Real, AMREX_SPACEDIM> prob_lo{}, prob_hi{}, dx{};
	[[nodiscard]] auto Domain() const -> Box const & { return domain; }
	[[nodiscard]] auto ProbLo(int d) const -> Real { return prob_lo[d]; }
	[[nodiscard]] auto CellSize(int d) const -> Real { return dx[d]; }
};

class Geometry
{
      public:
	Box domain;
	GpuArray<Real, AMREX_SPACEDIM> prob_lo{}, prob_hi{}, dx{};
	int periodic[3] = {0, 0, 0};
	[[nodiscard]] auto Domain() const -> Box const & { return domain; }
	[[nodiscard]] auto CellSizeArray() const -> GpuArray<Real, AMREX_SPACEDIM> { return dx; }
	[[nodiscard]] auto ProbLoArray() const -> GpuArray<Real, AMREX_SPACEDIM> { return prob_lo; }
	[[nodiscard]] auto ProbHiArray() const -> GpuArray<Real, AMREX_SPACEDIM> { return prob_hi; }
	[[nodiscard]] auto ProbLo(int d) const -> Real { return prob_lo[d]; }
	[[nodiscard]] auto CellSize(int d) const -> Real { return dx[d]; }
	[[nodiscard]] auto isPeriodic(int d) const -> bool { return periodic[d] != 0; }
	[[nodiscard]] auto isAllPeriodic() const -> bool
	{
		bool a = true;
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			a = a && isPeriodic(d);
		}
		return a;
	}
	[[nodiscard]] auto data() const -> GeometryData { return {domain, prob_lo, prob_hi, dx}; }
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
		std::istringstream s(it->second[0]);
		s >> val;
		return true;
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

#define QK_HOST_HIP(expr)                                                                                                                            \
	do {                                                                                                                                         \
		hipError_t e_ = (expr);                                                                                                              \
		if (e_ != hipSuccess) {                                                                                                              \
			amrex::Abort(std::string(#expr) + ": " + hipGetErrorString(e_));                                                             \
		}                                                                                                                                    \
	} while (0)

// FabArray<T> with device storage: one allocation for all boxes + the device table of Array4 descriptors
template <typename T> class FabArrayT
{
      public:
	FabArrayT() = default;
	FabArrayT(std::vector<Box> const &ba, int ncomp, int nghost, int facedir = -1) { define(ba, ncomp, nghost, facedir); }
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
		boxes_ = ba;
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
		QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_data_), sizeof(T) * std::max<Long>(total_, 1)));
		// fresh storage reads as zero (what the Python drivers' MultiFab(fill = 0) gives); QK_POISON=1 fills it with NaN bit patterns
		// instead, which makes any read of a cell that was never written visible in the results (debugging aid)
		QK_HOST_HIP(hipMemset(d_data_, std::getenv("QK_POISON") != nullptr ? 0xFF : 0, sizeof(T) * std::max<Long>(total_, 1)));
		for (size_t n = 0; n < ba.size(); ++n) {
			tab.emplace_back(d_data_ + offsets_[n], fabboxes_[n], ncomp);
		}
		QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_table_), sizeof(Array4<T>) * ba.size()));
		QK_HOST_HIP(hipMemcpy(d_table_, tab.data(), sizeof(Array4<T>) * ba.size(), hipMemcpyHostToDevice));
	}
	[[nodiscard]] auto size() const -> int { return static_cast<int>(boxes_.size()); }
	[[nodiscard]] auto nComp() const -> int { return ncomp_; }
	[[nodiscard]] auto nGrow() const -> int { return nghost_; }
	[[nodiscard]] auto boxArray() const -> std::vector<Box> const & { return boxes_; }
	[[nodiscard]] auto validbox(int b) const -> Box const & { return boxes_[b]; }
	[[nodiscard]] auto fabbox(int b) const -> Box const & { return fabboxes_[b]; }
	// host copy of one descriptor (device data pointer)
	[[nodiscard]] auto array(int b) const -> Array4<T> { return Array4<T>(d_data_ + offsets_[b], fabboxes_[b], ncomp_); }
	// device pointer to the descriptor table (MultiFab::arrays())
	[[nodiscard]] auto arrays() const -> Array4<T> * { return d_table_; }
	void setVal(T v)
	{
		std::vector<T> h(static_cast<size_t>(total_), v);
		QK_HOST_HIP(hipMemcpy(d_data_, h.data(), sizeof(T) * total_, hipMemcpyHostToDevice));
	}
	// host staging copies of one fab
	[[nodiscard]] auto copyToHost(int b) const -> std::vector<T>
	{
		std::vector<T> h(static_cast<size_t>(fabboxes_[b].numPts()) * ncomp_);
		QK_HOST_HIP(hipMemcpy(h.data(), d_data_ + offsets_[b], sizeof(T) * h.size(), hipMemcpyDeviceToHost));
		return h;
	}
	void copyFromHost(int b, std::vector<T> const &h) { QK_HOST_HIP(hipMemcpy(d_data_ + offsets_[b], h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice)); }
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
	// sum / norm over valid cells of component n (host reduction: diagnostics only, not on the timed path)
	[[nodiscard]] auto sum(int n) const -> double
	{
		// Kahan-compensated: AMReX reduces with a tree on the device, whose rounding error is far below a serial sum's
		double s = 0, comp = 0;
		for (int b = 0; b < size(); ++b) {
			auto h = copyToHost(b);
			Array4<T> a(h.data(), fabboxes_[b], ncomp_);
			ParallelFor(boxes_[b], [&](int i, int j, int k) {
				double const y = static_cast<double>(a(i, j, k, n)) - comp;
				double const t = s + y;
				comp = (t - s) - y;
				s = t;
			});
		}
		return s;
	}

      private:
	void release()
	{
		if (d_data_ != nullptr) {
			(void)hipFree(d_data_);
		}
		if (d_table_ != nullptr) {
			(void)hipFree(d_table_);
		}
		d_data_ = nullptr;
		d_table_ = nullptr;
	}
	std::vector<Box> boxes_, fabboxes_;
	std::vector<Long> offsets_;
	int ncomp_ = 0, nghost_ = 0, facedir_ = -1;
	Long total_ = 0;
	T *d_data_ = nullptr;
	Array4<T> *d_table_ = nullptr;
};
using MultiFab = FabArrayT<Real>;
using iMultiFab = FabArrayT<int>;
// amrex::TagBoxArray: one char per cell (amrex::TagBox::CLEAR = 0, BUF = 1, SET = 2)
using TagBoxArray = FabArrayT<char>;
struct TagBox {
	enum TagVal : char { CLEAR = 0, BUF = 1, SET = 2 };
};

} // namespace amrex

#endif // QK_HOST_AMREX_MINI_HPP_
