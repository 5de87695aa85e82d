// amr.hpp — CPU restatement of the data-parallel AMR pieces (test infrastructure, see oracle/__init__.py):
//   tagRelativeGradient   QuokkaSimulation<problem_t>::ErrorEst of src/problems/HydroBlast3D/test_hydro3d_blast.cpp:118-151
//                         (pressure, P > P_min) and src/problems/RadhydroShell/test_radhydro_shell.cpp:337-371 (density, rho >= rho_min)
//   averageDown           amrex::average_down / amrex_avgdown as called from AMRSimulation::AverageDownTo (src/simulation.hpp:1949-1964).
//                         AMReX is not vendored in /root/reference: the kernel is restated from its published form
//                         (crse = volfrac * sum over kref, jref, iref of the fine cells) — parity unpinned beyond that.
#ifndef ORACLE_AMR_HPP_
#define ORACLE_AMR_HPP_

#include <algorithm>
#include <cmath>

#include "grid.hpp"
#include "hydro.hpp"

namespace oracle
{

constexpr char TagBox_SET = 2; // amrex::TagBox::SET

// field < 0: HydroSystem::ComputePressure; otherwise the conserved component `field`
inline void tagRelativeGradient(HydroSystem const &hydro, Array4<const double> const &state, Array4<char> const &tag, Box const &box, int ndim, int field,
				double eta_threshold, double q_min, bool min_inclusive)
{
	auto q = [&](int i, int j, int k) -> double { return (field < 0) ? hydro.ComputePressure(state, i, j, k) : state(i, j, k, field); };
	for (int k = box.lo[2]; k <= box.hi[2]; ++k) {
		for (int j = box.lo[1]; j <= box.hi[1]; ++j) {
			for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
				double const P = q(i, j, k);
				double const del_x = std::max(std::abs(q(i + 1, j, k) - P), std::abs(P - q(i - 1, j, k)));
				double del = del_x;
				if (ndim >= 2) {
					double const del_y = std::max(std::abs(q(i, j + 1, k) - P), std::abs(P - q(i, j - 1, k)));
					del = std::max(del, del_y);
				}
				if (ndim == 3) {
					double const del_z = std::max(std::abs(q(i, j, k + 1) - P), std::abs(P - q(i, j, k - 1)));
					del = std::max(del, del_z);
				}
				double const gradient_indicator = del / P;
				bool const above = min_inclusive ? (P >= q_min) : (P > q_min);
				if ((gradient_indicator > eta_threshold) && above) {
					tag(i, j, k) = TagBox_SET;
				}
			}
		}
	}
}

// region: coarse cells to fill (must be covered by `fine`)
inline void averageDown(Array4<const double> const &fine, Array4<double> const &crse, Box const &region, int scomp, int ncomp, const int ratio[3])
{
	double const volfrac = 1.0 / static_cast<double>(ratio[0] * ratio[1] * ratio[2]);
	for (int n = 0; n < ncomp; ++n) {
		for (int k = region.lo[2]; k <= region.hi[2]; ++k) {
			for (int j = region.lo[1]; j <= region.hi[1]; ++j) {
				for (int i = region.lo[0]; i <= region.hi[0]; ++i) {
					double c = 0.0;
					for (int kref = 0; kref < ratio[2]; ++kref) {
						for (int jref = 0; jref < ratio[1]; ++jref) {
							for (int iref = 0; iref < ratio[0]; ++iref) {
								c += fine(i * ratio[0] + iref, j * ratio[1] + jref, k * ratio[2] + kref, n + scomp);
							}
						}
					}
					crse(i, j, k, n + scomp) = volfrac * c;
				}
			}
		}
	}
}

} // namespace oracle

#endif
