// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the reference algorithm (quokka-astro/quokka @ 2024-10-24).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
// under oracle/. The product path (quokka_amd/, include/) never links or calls this.
//
// grid.hpp: minimal stand-ins for the AMReX containers the reference's hot path uses
// (Box, Array4, FArrayBox, MultiFab, Geometry, BCRec). AMReX itself is an un-vendored
// submodule of the reference (extern/amrex is empty), so the semantics restated here are
// the documented AMReX ones: Fortran-order Array4 with the component index outermost,
// inclusive Box bounds, ghost cells only in the active dimensions (AMREX_SPACEDIM).
#ifndef ORACLE_GRID_HPP_
#define ORACLE_GRID_HPP_

#include <algorithm>
#include <array>
#include <cassert>
#include <cstdint>
#include <vector>

namespace oracle
{

struct Box {
	int lo[3] = {0, 0, 0};
	int hi[3] = {0, 0, 0}; // inclusive
	[[nodiscard]] auto length(int d) const -> int { return hi[d] - lo[d] + 1; }
	[[nodiscard]] auto numPts() const -> int64_t { return static_cast<int64_t>(length(0)) * length(1) * length(2); }
	[[nodiscard]] auto contains(int i, int j, int k) const -> bool
	{
		return i >= lo[0] && i <= hi[0] && j >= lo[1] && j <= hi[1] && k >= lo[2] && k <= hi[2];
	}
	[[nodiscard]] auto ok() const -> bool { return hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2]; }
};

// grow by ng cells in the first ndim dimensions only (amrex::grow in an ndim-dimensional build)
inline auto grow(Box b, int ng, int ndim) -> Box
{
	for (int d = 0; d < ndim; ++d) {
		b.lo[d] -= ng;
		b.hi[d] += ng;
	}
	return b;
}

// amrex::convert(box, IntVect::TheDimensionVector(dir)): cell box -> face (nodal in dir) box
inline auto faceBox(Box b, int dir) -> Box
{
	b.hi[dir] += 1;
	return b;
}

inline auto intersect(Box const &a, Box const &b) -> Box
{
	Box r;
	for (int d = 0; d < 3; ++d) {
		r.lo[d] = std::max(a.lo[d], b.lo[d]);
		r.hi[d] = std::min(a.hi[d], b.hi[d]);
	}
	return r;
}

inline auto shift(Box b, int const s[3]) -> Box
{
	for (int d = 0; d < 3; ++d) {
		b.lo[d] += s[d];
		b.hi[d] += s[d];
	}
	return b;
}

// amrex::Array4<T>: p[(i-lo.x) + jstride*(j-lo.y) + kstride*(k-lo.z) + nstride*n]
template <typename T> struct Array4 {
	T *p = nullptr;
	int lo[3] = {0, 0, 0};
	int hi[3] = {0, 0, 0};
	int64_t jstride = 0, kstride = 0, nstride = 0;
	int ncomp = 0;

	Array4() = default;
	Array4(T *ptr, Box const &bx, int nc) : p(ptr), ncomp(nc)
	{
		for (int d = 0; d < 3; ++d) {
			lo[d] = bx.lo[d];
			hi[d] = bx.hi[d];
		}
		jstride = bx.length(0);
		kstride = jstride * bx.length(1);
		nstride = kstride * bx.length(2);
	}
	auto operator()(int i, int j, int k, int n = 0) const -> T &
	{
#ifdef ORACLE_BOUNDS_CHECK
		assert(i >= lo[0] && i <= hi[0] && j >= lo[1] && j <= hi[1] && k >= lo[2] && k <= hi[2] && n >= 0 && n < ncomp);
#endif
		return p[(i - lo[0]) + jstride * (j - lo[1]) + kstride * (k - lo[2]) + nstride * n];
	}
	[[nodiscard]] auto contains(int i, int j, int k) const -> bool
	{
		return i >= lo[0] && i <= hi[0] && j >= lo[1] && j <= hi[1] && k >= lo[2] && k <= hi[2];
	}
	[[nodiscard]] auto box() const -> Box
	{
		Box b;
		for (int d = 0; d < 3; ++d) {
			b.lo[d] = lo[d];
			b.hi[d] = hi[d];
		}
		return b;
	}
};

template <typename T> struct Fab {
	Box bx;
	int nc = 0;
	std::vector<T> d;
	Fab() = default;
	Fab(Box const &b, int ncomp, T init = T(0)) : bx(b), nc(ncomp), d(static_cast<size_t>(b.numPts()) * ncomp, init) {}
	auto array() -> Array4<T> { return Array4<T>(d.data(), bx, nc); }
	auto const_array() const -> Array4<const T> { return Array4<const T>(d.data(), bx, nc); }
	void setVal(T v) { std::fill(d.begin(), d.end(), v); }
};

// one level's worth of boxes: valid boxes + ng ghost cells (in the active dims)
template <typename T> struct MultiFabT {
	std::vector<Box> valid; // cell-centred valid boxes
	int ng = 0;
	int nc = 0;
	int ndim = 3;
	int facedir = -1; // -1: cell-centred; 0..2: nodal in that direction
	std::vector<Fab<T>> fabs;

	MultiFabT() = default;
	MultiFabT(std::vector<Box> const &ba, int ncomp, int nghost, int ndim_in, int facedir_in = -1, T init = T(0))
	    : valid(ba), ng(nghost), nc(ncomp), ndim(ndim_in), facedir(facedir_in)
	{
		// one thread per box allocates AND first-touches its fab (cpu_baseline leg of bench.py: the reference's
		// ~130 temporaries per step are otherwise initialised by one core)
		fabs.resize(ba.size());
		const int n = static_cast<int>(ba.size());
		_Pragma("omp parallel for schedule(static)")
		for (int b = 0; b < n; ++b) {
			Box fb = (facedir >= 0) ? faceBox(ba[b], facedir) : ba[b];
			fabs[b] = Fab<T>(grow(fb, ng, ndim), nc, init);
		}
	}
	[[nodiscard]] auto size() const -> int { return static_cast<int>(valid.size()); }
	// validbox of the MFIter (for face MultiFabs this is the nodal box)
	[[nodiscard]] auto validbox(int b) const -> Box { return (facedir >= 0) ? faceBox(valid[b], facedir) : valid[b]; }
	auto array(int b) -> Array4<T> { return fabs[b].array(); }
	auto const_array(int b) const -> Array4<const T> { return fabs[b].const_array(); }
	void setVal(T v)
	{
		const int n = static_cast<int>(fabs.size());
		_Pragma("omp parallel for schedule(static)")
		for (int b = 0; b < n; ++b) {
			fabs[b].setVal(v);
		}
	}
};

using MultiFab = MultiFabT<double>;
using iMultiFab = MultiFabT<int>;

// amrex::BCType values (AMReX_BC_TYPES.H)
enum BCType : int { reflect_odd = -1, int_dir = 0, reflect_even = 1, foextrap = 2, ext_dir = 3 };

struct BCRec {
	int lo[3] = {int_dir, int_dir, int_dir};
	int hi[3] = {int_dir, int_dir, int_dir};
};

struct Geometry {
	Box domain;
	double prob_lo[3] = {0, 0, 0};
	double prob_hi[3] = {1, 1, 1};
	double dx[3] = {1, 1, 1};
	int periodic[3] = {0, 0, 0};
	int ndim = 3;
	[[nodiscard]] auto isAllPeriodic() const -> bool
	{
		bool all = true;
		for (int d = 0; d < ndim; ++d) {
			all = all && (periodic[d] != 0);
		}
		return all;
	}
};

// BoxArray(domain).maxSize(max_grid_size) for domains that are multiples of max_grid_size
inline auto chopDomain(Box const &domain, int const max_grid_size[3]) -> std::vector<Box>
{
	std::vector<Box> ba;
	int nb[3];
	for (int d = 0; d < 3; ++d) {
		int const len = domain.length(d);
		nb[d] = (len + max_grid_size[d] - 1) / max_grid_size[d];
	}
	for (int kb = 0; kb < nb[2]; ++kb) {
		for (int jb = 0; jb < nb[1]; ++jb) {
			for (int ib = 0; ib < nb[0]; ++ib) {
				int const idx[3] = {ib, jb, kb};
				Box b;
				for (int d = 0; d < 3; ++d) {
					// AMReX chops evenly: sizes differ by at most 1 block; for the configs here
					// the domain is an exact multiple, so every chunk has equal length.
					int const len = domain.length(d);
					int const base = len / nb[d];
					int const rem = len % nb[d];
					int const start = idx[d] * base + std::min(idx[d], rem);
					int const sz = base + (idx[d] < rem ? 1 : 0);
					b.lo[d] = domain.lo[d] + start;
					b.hi[d] = b.lo[d] + sz - 1;
				}
				ba.push_back(b);
			}
		}
	}
	return ba;
}

} // namespace oracle

#endif // ORACLE_GRID_HPP_
