// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// boundary.hpp: restatement of the level-0 branch of AMRSimulation::fillBoundaryConditions
// (reference src/simulation.hpp:1751-1776):
//   1. state.FillBoundary(geom.periodicity())   — AMReX: every ghost cell whose (periodically
//      shifted) index lies in another box's VALID region is overwritten with that value
//      (faces + edges + corners, cross=false);
//   2. if not all-periodic: PhysBCFunct(GpuBndryFuncFab<setBoundaryFunctor>) — AMReX applies, to
//      every ghost cell outside the (periodically grown) domain, first FilccCell (reflect_even /
//      reflect_odd / foextrap per component, dimension by dimension; launched faces -> edges ->
//      corners so that multi-dimensional ghosts compose: AMReX_FilCC_3D_C.H, AMReX_PhysBCFunct.H),
//      then the user functor AMRSimulation::setCustomBoundaryConditions (simulation.hpp:1550-1561).
// AMReX is un-vendored (extern/amrex empty): the formulas are the published AMReX ones
// (reflect_even: q(lo-1-m) = q(lo+m); reflect_odd: negated; foextrap: q(lo)).
#ifndef ORACLE_BOUNDARY_HPP_
#define ORACLE_BOUNDARY_HPP_

#include <functional>
#include <vector>

#include "grid.hpp"

namespace oracle
{

// user hook: (i,j,k, array over the fab, domain box, time)
using CustomBCFunc = std::function<void(int, int, int, Array4<double> const &, Box const &, double)>;

// MultiFab::FillBoundary(periodicity) for cell-centred data
template <typename T> inline void FillBoundary(MultiFabT<T> &mf, Geometry const &geom)
{
	int const nb = mf.size();
	// enumerate periodic shifts
	std::vector<std::array<int, 3>> shifts;
	int rng[3] = {0, 0, 0};
	for (int d = 0; d < geom.ndim; ++d) {
		rng[d] = (geom.periodic[d] != 0) ? 1 : 0;
	}
	for (int sz = -rng[2]; sz <= rng[2]; ++sz) {
		for (int sy = -rng[1]; sy <= rng[1]; ++sy) {
			for (int sx = -rng[0]; sx <= rng[0]; ++sx) {
				shifts.push_back({sx * geom.domain.length(0), sy * geom.domain.length(1), sz * geom.domain.length(2)});
			}
		}
	}
	// ghost cells of `dst` are written from VALID cells of the sources: destinations are independent
	_Pragma("omp parallel for schedule(dynamic)")
	for (int dst = 0; dst < nb; ++dst) {
		auto darr = mf.array(dst);
		Box const dbox = mf.fabs[dst].bx;
		for (int src = 0; src < nb; ++src) {
			auto sarr = mf.const_array(src);
			for (auto const &s : shifts) {
				bool const zero_shift = (s[0] == 0 && s[1] == 0 && s[2] == 0);
				if (src == dst && zero_shift) {
					continue;
				}
				int const sh[3] = {s[0], s[1], s[2]};
				// source valid box, shifted into the destination's index space
				Box const sv = shift(mf.valid[src], sh);
				Box const isect = intersect(dbox, sv);
				if (!isect.ok()) {
					continue;
				}
				// never overwrite valid cells: boxes of a level do not overlap and a periodic image lies outside the domain, so the
				// intersection holds none of the destination's valid cells — tested once per intersection, per cell only if it does
				bool const touches_valid = intersect(isect, mf.valid[dst]).ok();
				for (int n = 0; n < mf.nc; ++n) {
					for (int k = isect.lo[2]; k <= isect.hi[2]; ++k) {
						for (int j = isect.lo[1]; j <= isect.hi[1]; ++j) {
							if (!touches_valid) {
								T *d = &darr(isect.lo[0], j, k, n);
								T const *sp = &sarr(isect.lo[0] - sh[0], j - sh[1], k - sh[2], n);
								int const len = isect.hi[0] - isect.lo[0] + 1;
								for (int e = 0; e < len; ++e) {
									d[e] = sp[e];
								}
								continue;
							}
							for (int i = isect.lo[0]; i <= isect.hi[0]; ++i) {
								if (mf.valid[dst].contains(i, j, k)) {
									continue;
								}
								darr(i, j, k, n) = sarr(i - sh[0], j - sh[1], k - sh[2], n);
							}
						}
					}
				}
			}
		}
	}
}

// PhysBCFunct: FilccCell + user functor on every ghost cell outside the domain in a
// non-periodic dimension.  Composition across dimensions = mirror/clamp the index in each
// out-of-domain non-periodic dimension, sign = product of reflect_odd factors.
inline void FillPhysicalBoundary(MultiFab &mf, Geometry const &geom, std::vector<BCRec> const &bcs, CustomBCFunc const &userFunc, double time)
{
	Box const &dom = geom.domain;
	_Pragma("omp parallel for schedule(dynamic)")
	for (int b = 0; b < mf.size(); ++b) {
		auto arr = mf.array(b);
		Box const fb = mf.fabs[b].bx;
		// process in the AMReX launch order: cells outside in exactly 1 dim, then 2, then 3,
		// so that sources of edge/corner cells are already filled.  Only cells outside the domain are visited.
		auto sideOf = [&](int d, int v) -> int { // -1 below lo, +1 above hi
			if (d >= geom.ndim || geom.periodic[d] != 0) {
				return 0;
			}
			return (v < dom.lo[d]) ? -1 : ((v > dom.hi[d]) ? 1 : 0);
		};
		bool outside = false; // a fab inside the domain in every non-periodic dimension has no such cell
		for (int d = 0; d < geom.ndim; ++d) {
			outside = outside || sideOf(d, fb.lo[d]) != 0 || sideOf(d, fb.hi[d]) != 0;
		}
		if (!outside) {
			continue;
		}
		auto cell = [&](int i, int j, int k, int const side[3]) {
			int const idx[3] = {i, j, k};
			// FilccCell: dimension by dimension, x then y then z (AMReX_FilCC_3D_C.H)
			for (int n = 0; n < mf.nc; ++n) {
				BCRec const &bc = bcs[n];
				for (int d = 0; d < geom.ndim; ++d) {
					if (side[d] == 0) {
						continue;
					}
					int src[3] = {i, j, k};
					int const type = (side[d] < 0) ? bc.lo[d] : bc.hi[d];
					double sgnf = 1.0;
					bool apply = true;
					if (side[d] < 0) {
						int const ilo = dom.lo[d];
						if (type == foextrap) {
							src[d] = ilo;
						} else if (type == reflect_even) {
							src[d] = 2 * ilo - idx[d] - 1;
						} else if (type == reflect_odd) {
							src[d] = 2 * ilo - idx[d] - 1;
							sgnf = -1.0;
						} else {
							apply = false; // ext_dir / int_dir: untouched by FilccCell
						}
					} else {
						int const ihi = dom.hi[d];
						if (type == foextrap) {
							src[d] = ihi;
						} else if (type == reflect_even) {
							src[d] = 2 * ihi - idx[d] + 1;
						} else if (type == reflect_odd) {
							src[d] = 2 * ihi - idx[d] + 1;
							sgnf = -1.0;
						} else {
							apply = false;
						}
					}
					if (apply) {
						double const v = arr(src[0], src[1], src[2], n);
						arr(i, j, k, n) = (sgnf < 0) ? -v : v;
					}
				}
			}
			// user functor is invoked for every such cell (simulation.hpp:1712-1720)
			if (userFunc) {
				userFunc(i, j, k, arr, dom, time);
			}
		};
		bool const xwall = geom.periodic[0] == 0;
		for (int pass = 1; pass <= geom.ndim; ++pass) {
			for (int k = fb.lo[2]; k <= fb.hi[2]; ++k) {
				for (int j = fb.lo[1]; j <= fb.hi[1]; ++j) {
					int side[3] = {0, sideOf(1, j), sideOf(2, k)};
					int const nout_jk = (side[1] != 0 ? 1 : 0) + (side[2] != 0 ? 1 : 0);
					// the row in three segments: below the domain in x, inside, above (x periodic: one segment)
					int const seg_lo[3] = {fb.lo[0], xwall ? std::max(fb.lo[0], dom.lo[0]) : fb.lo[0], std::max(fb.lo[0], dom.hi[0] + 1)};
					int const seg_hi[3] = {std::min(fb.hi[0], dom.lo[0] - 1), xwall ? std::min(fb.hi[0], dom.hi[0]) : fb.hi[0], fb.hi[0]};
					int const seg_side[3] = {-1, 0, 1};
					for (int sgm = 0; sgm < 3; ++sgm) {
						if (!xwall && sgm != 1) {
							continue;
						}
						if (nout_jk + (seg_side[sgm] != 0 ? 1 : 0) != pass) {
							continue;
						}
						side[0] = seg_side[sgm];
						for (int i = seg_lo[sgm]; i <= seg_hi[sgm]; ++i) {
							cell(i, j, k, side);
						}
					}
				}
			}
		}
	}
}

// AMRSimulation::fillBoundaryConditions, level-0 branch (simulation.hpp:1751-1776)
inline void fillBoundaryConditions(MultiFab &state, Geometry const &geom, std::vector<BCRec> const &bcs, CustomBCFunc const &userFunc, double time)
{
	FillBoundary(state, geom);
	if (!geom.isAllPeriodic()) {
		FillPhysicalBoundary(state, geom, bcs, userFunc, time);
	}
}

} // namespace oracle

#endif // ORACLE_BOUNDARY_HPP_
