// util_compat.hpp — the reference's small utilities that its problem files include next to the simulation headers:
//   util/fextract.hpp   fextract(mf, geom, idir, slice_coord, center): the 1-D profile of every component along direction idir
//                       (reference src/util/fextract.cpp:15-190), used by 43 problem files to compare with exact solutions;
//   util/ArrayUtil.hpp  strided_vector_from;
//   util/valarray.hpp   quokka::valarray<T, d>, element-wise arithmetic on small fixed-size arrays (reference src/util/valarray.hpp:23-294).
#ifndef QK_HOST_UTIL_COMPAT_HPP_
#define QK_HOST_UTIL_COMPAT_HPP_

#include <cmath>
#include <initializer_list>
#include <tuple>
#include <vector>

#include "../amrex_mini.hpp"

template <typename T> auto strided_vector_from(std::vector<T> &v, int stride) -> std::vector<T>
{
	std::vector<T> out;
	for (std::size_t i = 0; i < v.size(); i += stride) {
		out.push_back(v[i]);
	}
	return out;
}

// positions (cell centres) along idir and, per component, the values on the line through the lower-left (or central) transverse cell
inline auto fextract(amrex::MultiFab &mf, amrex::Geometry &geom, int idir, amrex::Real /*slice_coord*/, bool center = false)
    -> std::tuple<amrex::Vector<amrex::Real>, amrex::Vector<amrex::Gpu::HostVector<amrex::Real>>>
{
	auto const &dom = geom.Domain();
	if (idir < 0 || idir >= AMREX_SPACEDIM) {
		amrex::Abort("invalid direction!");
	}
	int loc[3] = {dom.lo[0], dom.lo[1], dom.lo[2]};
	if (center) {
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			loc[d] = (dom.hi[d] - dom.lo[d] + 1) / 2 + dom.lo[d];
		}
	}
	int const n = dom.length(idir);
	amrex::Vector<amrex::Real> pos(n);
	for (int i = 0; i < n; ++i) {
		pos[i] = geom.ProbLo(idir) + (dom.lo[idir] + i + 0.5) * geom.CellSize(idir);
	}
	amrex::Vector<amrex::Gpu::HostVector<amrex::Real>> data(mf.nComp());
	for (auto &v : data) {
		v.resize(n);
	}
	for (int b = 0; b < mf.size(); ++b) {
		amrex::Box const &vb = mf.validbox(b);
		bool on = true;
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			on = on && (d == idir || (loc[d] >= vb.lo[d] && loc[d] <= vb.hi[d]));
		}
		if (!on) {
			continue;
		}
		auto h = mf.copyToHost(b);
		amrex::Array4<amrex::Real> a(h.data(), mf.fabbox(b), mf.nComp());
		for (int i = vb.lo[idir]; i <= vb.hi[idir]; ++i) {
			int c[3] = {loc[0], loc[1], loc[2]};
			c[idir] = i;
			for (int ivar = 0; ivar < mf.nComp(); ++ivar) {
				data[ivar][i - dom.lo[idir]] = a(c[0], c[1], c[2], ivar);
			}
		}
	}
	return {pos, data};
}

namespace quokka
{
template <typename T, int d> class valarray
{
      public:
	QK_HD valarray() = default;
	// an initialiser list shorter than d zero-fills the tail (reference src/util/valarray.hpp:41-46; HLLC relies on it)
	QK_HD valarray(std::initializer_list<T> list)
	{
		int n = 0;
		for (auto const &v : list) {
			if (n < d) {
				values[n++] = v;
			}
		}
		for (; n < d; ++n) {
			values[n] = T{};
		}
	}
	QK_HD auto operator[](int i) -> T & { return values[i]; }
	QK_HD auto operator[](int i) const -> T const & { return values[i]; }
	[[nodiscard]] QK_HD constexpr auto size() const -> int { return d; }
	QK_HD void fillin(T v)
	{
		for (int i = 0; i < d; ++i) {
			values[i] = v;
		}
	}
	[[nodiscard]] QK_HD auto hasnan() const -> bool
	{
		for (int i = 0; i < d; ++i) {
			if (values[i] != values[i]) {
				return true;
			}
		}
		return false;
	}
	QK_HD auto operator+=(valarray const &o) -> valarray &
	{
		for (int i = 0; i < d; ++i) {
			values[i] += o.values[i];
		}
		return *this;
	}
	QK_HD auto operator-=(valarray const &o) -> valarray &
	{
		for (int i = 0; i < d; ++i) {
			values[i] -= o.values[i];
		}
		return *this;
	}
	QK_HD auto operator*=(T s) -> valarray &
	{
		for (int i = 0; i < d; ++i) {
			values[i] *= s;
		}
		return *this;
	}
	QK_HD auto operator/=(T s) -> valarray &
	{
		for (int i = 0; i < d; ++i) {
			values[i] /= s;
		}
		return *this;
	}

      private:
	T values[d > 0 ? d : 1]{};
};
#define QK_VALARRAY_BINOP(OP)                                                                                                                        \
	template <typename T, int d> QK_HD auto operator OP(valarray<T, d> const &a, valarray<T, d> const &b) -> valarray<T, d>                       \
	{                                                                                                                                            \
		valarray<T, d> r;                                                                                                                    \
		for (int i = 0; i < d; ++i) {                                                                                                        \
			r[i] = a[i] OP b[i];                                                                                                         \
		}                                                                                                                                    \
		return r;                                                                                                                            \
	}                                                                                                                                            \
	template <typename T, int d> QK_HD auto operator OP(valarray<T, d> const &a, T const &s) -> valarray<T, d>                                    \
	{                                                                                                                                            \
		valarray<T, d> r;                                                                                                                    \
		for (int i = 0; i < d; ++i) {                                                                                                        \
			r[i] = a[i] OP s;                                                                                                            \
		}                                                                                                                                    \
		return r;                                                                                                                            \
	}                                                                                                                                            \
	template <typename T, int d> QK_HD auto operator OP(T const &s, valarray<T, d> const &a) -> valarray<T, d>                                    \
	{                                                                                                                                            \
		valarray<T, d> r;                                                                                                                    \
		for (int i = 0; i < d; ++i) {                                                                                                        \
			r[i] = s OP a[i];                                                                                                            \
		}                                                                                                                                    \
		return r;                                                                                                                            \
	}
QK_VALARRAY_BINOP(+)
QK_VALARRAY_BINOP(-)
QK_VALARRAY_BINOP(*)
QK_VALARRAY_BINOP(/)
#undef QK_VALARRAY_BINOP
template <typename T, int d> QK_HD auto sum(valarray<T, d> const &a) -> T
{
	T s{};
	for (int i = 0; i < d; ++i) {
		s += a[i];
	}
	return s;
}
template <typename T, int d> QK_HD auto abs(valarray<T, d> const &a) -> valarray<T, d>
{
	valarray<T, d> r;
	for (int i = 0; i < d; ++i) {
		r[i] = std::abs(a[i]);
	}
	return r;
}
template <typename T, int d> QK_HD auto min(valarray<T, d> const &a) -> T
{
	T m = a[0];
	for (int i = 1; i < d; ++i) {
		m = (a[i] < m) ? a[i] : m;
	}
	return m;
}
template <typename T, int d> QK_HD auto isnan(valarray<T, d> const &a) -> bool { return a.hasnan(); }
} // namespace quokka

#endif // QK_HOST_UTIL_COMPAT_HPP_
