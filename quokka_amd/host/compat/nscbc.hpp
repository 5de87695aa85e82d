// nscbc.hpp of the host mirror: Navier-Stokes characteristic boundary conditions for the ghost cells of subsonic in- and outflows — what the
// reference's `hydro/NSCBC_inflow.hpp` and `hydro/NSCBC_outflow.hpp` give a problem's setCustomBoundaryConditions (same names, arguments and
// behaviour: `NSCBC::setInflowX1Lower`, `setInflowX1LowerLowOrder`, `setOutflowBoundary<problem_t, DIR, SIDE>`, `setOutflowBoundaryLowOrder`).
// These run inside the boundary kernel the host mirror launches from the problem's own functor (device mode).
//
// Formulation.  With the wave amplitudes per unit speed of the 1-D Euler equations along the boundary normal n,
//     A- = dP/dn - rho c du/dn   (speed u - c),      A+ = dP/dn + rho c du/dn   (speed u + c),      S = c^2 drho/dn - dP/dn   (speed u),
// the normal derivative of the primitive state is
//     dP/dn = (A+ + A-) / 2,     du/dn = (A+ - A-) / (2 rho c),     drho/dn = (S + (A+ + A-) / 2) / c^2,
// and v, w, the auxiliary internal energy and the passive scalars are advected at speed u.  Amplitudes of waves LEAVING the domain are taken from
// one-sided differences of the interior; amplitudes of waves ENTERING are modelled (Poinsot & Lele 1992; Yoo & Im 2007 for the transverse and
// relaxation terms):
//   outflow:  incoming acoustic amplitude = [K (P - P_t) + (beta - 1) T] / (speed),  K = c (1 - M^2) / (4 L),  beta = M,  T = transverse terms;
//   inflow :  incoming acoustic amplitude relaxes u to u_t  (eta_5 rho c^2 (1 - M^2) (u - u_t) / L),  the entropy amplitude relaxes the
//             temperature to T_t  (eta_2 rho R c (T - T_t) / L),  advected quantities relax to their targets  (eta c (q - q_t) / (L u)).
// The ghost cells continue the boundary cell with the cubic through (two interior cells, the boundary cell, this derivative) — reference
// src/hydro/NSCBC_outflow.hpp:232-357, NSCBC_inflow.hpp:99-150 for the ghost-cell rule and the low-order variants; the characteristic form above is
// algebraically the reference's SymPy-generated expressions (:63-96 resp. :60-95), re-derived.
#ifndef QK_HOST_NSCBC_HPP_
#define QK_HOST_NSCBC_HPP_

#include <cmath>

#include "../quokka_host.hpp"

namespace NSCBC
{
enum class BoundarySide { Lower, Upper };

namespace detail
{
template <typename problem_t> using PrimVec = quokka::valarray<amrex::Real, HydroSystem<problem_t>::nvar_>;

// velocity components in the frame of the boundary normal DIR: (normal, next axis, the one after) — and back
template <typename problem_t, FluxDir DIR> AMREX_GPU_DEVICE AMREX_FORCE_INLINE auto toNormalFrame(PrimVec<problem_t> const &q) -> PrimVec<problem_t>
{
	constexpr int d = static_cast<int>(DIR);
	PrimVec<problem_t> r = q;
	r[1] = q[1 + d];
	r[2] = q[1 + (d + 1) % 3];
	r[3] = q[1 + (d + 2) % 3];
	return r;
}
template <typename problem_t, FluxDir DIR> AMREX_GPU_DEVICE AMREX_FORCE_INLINE auto fromNormalFrame(PrimVec<problem_t> const &q) -> PrimVec<problem_t>
{
	constexpr int d = static_cast<int>(DIR);
	PrimVec<problem_t> r = q;
	r[1 + d] = q[1];
	r[1 + (d + 1) % 3] = q[2];
	r[1 + (d + 2) % 3] = q[3];
	return r;
}

template <typename problem_t> AMREX_GPU_DEVICE AMREX_FORCE_INLINE auto primAt(amrex::Array4<amrex::Real> const &consVar, int const idx[3]) -> PrimVec<problem_t>
{
	return HydroSystem<problem_t>::ComputePrimVars(consVar, idx[0], idx[1], idx[2]);
}

// centred difference along `axis` through the cell `at` (zero when a neighbour is outside the array)
template <typename problem_t>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE auto centredDifference(amrex::Array4<amrex::Real> const &consVar, int const at[3], int axis, amrex::Real h) -> PrimVec<problem_t>
{
	int p[3] = {at[0], at[1], at[2]}, m[3] = {at[0], at[1], at[2]};
	p[axis] += 1;
	m[axis] -= 1;
	PrimVec<problem_t> d{};
	if (consVar.contains(p[0], p[1], p[2]) && consVar.contains(m[0], m[1], m[2])) {
		d = (primAt<problem_t>(consVar, p) - primAt<problem_t>(consVar, m)) / (2.0 * h);
	}
	return d;
}

// transverse derivatives at the boundary cell, in the order (next axis, the one after).  The reference's x-boundary routine in a 3-D build stores
// the z derivative in the slot of the y derivative and leaves the z slot zero (NSCBC_outflow.hpp:121-127); kept, so that results agree.
template <typename problem_t, FluxDir DIR>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE void transverseDerivatives(amrex::Array4<amrex::Real> const &consVar, int const at[3], amrex::GeometryData const &geom,
							       PrimVec<problem_t> &d1, PrimVec<problem_t> &d2)
{
	constexpr int d = static_cast<int>(DIR);
	constexpr int a1 = (d + 1) % 3, a2 = (d + 2) % 3;
	d1 = PrimVec<problem_t>{};
	d2 = PrimVec<problem_t>{};
	if (a1 < AMREX_SPACEDIM) {
		d1 = centredDifference<problem_t>(consVar, at, a1, geom.CellSize(a1));
	}
	if (a2 < AMREX_SPACEDIM) {
		d2 = centredDifference<problem_t>(consVar, at, a2, geom.CellSize(a2));
	}
	if constexpr (DIR == FluxDir::X1 && AMREX_SPACEDIM == 3) {
		d1 = d2;
		d2 = PrimVec<problem_t>{};
	}
}

// dQ/dn of a subsonic outflow; Q, dQ_dn (one-sided, from the interior) and the transverse derivatives dQ_t1, dQ_t2 in the normal frame
template <typename problem_t, BoundarySide SIDE>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE auto outflowNormalDerivative(PrimVec<problem_t> const &Q, PrimVec<problem_t> const &dQ_dn, PrimVec<problem_t> const &dQ_t1,
								 PrimVec<problem_t> const &dQ_t2, amrex::Real P_t, amrex::Real L) -> PrimVec<problem_t>
{
	const amrex::Real rho = Q[0], u = Q[1], v = Q[2], w = Q[3], P = Q[4];
	const amrex::Real c = quokka::EOS<problem_t>::ComputeSoundSpeed(rho, P);
	const amrex::Real M = std::clamp(std::sqrt(u * u + v * v + w * w) / c, 0., 1.);
	const amrex::Real beta = M;
	const amrex::Real K = 0.25 * c * (1 - M * M) / L;
	const amrex::Real rc = rho * c;
	const amrex::Real rcc = rho * (c * c);
	// amplitudes carried by the interior data
	const amrex::Real Aminus_data = dQ_dn[4] - rc * dQ_dn[1];
	const amrex::Real Aplus_data = dQ_dn[4] + rc * dQ_dn[1];
	const amrex::Real S = (c * c) * dQ_dn[0] - dQ_dn[4];
	amrex::Real Aminus = Aminus_data, Aplus = Aplus_data;
	if (SIDE == BoundarySide::Upper) { // the u - c wave enters
		const amrex::Real T = dQ_t1[4] * v + dQ_t2[4] * w - dQ_t1[1] * v * rc - dQ_t2[1] * w * rc + dQ_t1[2] * rcc + dQ_t2[3] * rcc;
		Aminus = (K * (P - P_t) + (beta - 1) * T) / (u - c);
	} else { // the u + c wave enters
		const amrex::Real T = dQ_t1[4] * v + dQ_t2[4] * w + dQ_t1[1] * v * rc + dQ_t2[1] * w * rc + dQ_t1[2] * rcc + dQ_t2[3] * rcc;
		Aplus = (K * (P - P_t) + (beta - 1) * T) / (u + c);
	}
	PrimVec<problem_t> d = dQ_dn; // v, w, the auxiliary energy and the scalars leave with the flow: interior data
	d[4] = 0.5 * (Aplus + Aminus);
	d[1] = 0.5 * (Aplus - Aminus) / rc;
	d[0] = (S + 0.5 * (Aplus + Aminus)) / (c * c);
	return d;
}

// the cell `steps` cells beyond the boundary cell (1..4) of the cubic continuation; Q_b boundary cell, Q_in its interior neighbour, g = h dQ/dn
// signed towards the ghost cells
template <typename problem_t>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE auto continuation(PrimVec<problem_t> const &Q_in, PrimVec<problem_t> const &Q_b, PrimVec<problem_t> const &g, int steps)
    -> PrimVec<problem_t>
{
	const PrimVec<problem_t> Q1 = Q_in + 2.0 * g;
	if (steps == 1) {
		return Q1;
	}
	const PrimVec<problem_t> Q2 = -2.0 * Q_in - 3.0 * Q_b + 6.0 * Q1 - 6.0 * g;
	if (steps == 2) {
		return Q2;
	}
	const PrimVec<problem_t> Q3 = 3.0 * Q_in + 10.0 * Q_b - 18.0 * Q1 + 6.0 * Q2 + 12.0 * g;
	if (steps == 3) {
		return Q3;
	}
	return -2.0 * Q_in - 13.0 * Q_b + 24.0 * Q1 - 12.0 * Q2 + 4.0 * Q3 - 12.0 * g;
}

template <typename problem_t>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE void storeGhost(amrex::Array4<amrex::Real> const &consVar, int i, int j, int k, PrimVec<problem_t> const &prim)
{
	const PrimVec<problem_t> cons = HydroSystem<problem_t>::ComputeConsVars(prim);
	for (int n = 0; n < HydroSystem<problem_t>::nvar_; ++n) {
		consVar(i, j, k, n) = cons[n];
	}
}
} // namespace detail

// subsonic (or supersonic) outflow through the SIDE face normal to DIR, far-field pressure P_outflow
template <typename problem_t, FluxDir DIR, BoundarySide SIDE>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE void setOutflowBoundary(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar, amrex::GeometryData const &geom,
							    const amrex::Real P_outflow)
{
	using Vec = detail::PrimVec<problem_t>;
	constexpr int d = static_cast<int>(DIR);
	constexpr int inward = (SIDE == BoundarySide::Lower) ? 1 : -1;
	auto [i, j, k] = iv.dim3();
	const int here[3] = {i, j, k};
	const auto edge = (SIDE == BoundarySide::Lower) ? geom.Domain().loVect3d() : geom.Domain().hiVect3d();
	int b0[3] = {i, j, k}, b1[3] = {i, j, k}, b2[3] = {i, j, k};
	b0[d] = edge[d];
	b1[d] = edge[d] + inward;
	b2[d] = edge[d] + 2 * inward;
	const amrex::Real h = geom.CellSize(d);
	const Vec Q_b = detail::primAt<problem_t>(consVar, b0);
	const Vec Q_1 = detail::primAt<problem_t>(consVar, b1);
	const Vec Q_2 = detail::primAt<problem_t>(consVar, b2);
	// second-order one-sided derivative along +n
	Vec dQ_dn = (Q_2 - 4.0 * Q_1 + 3.0 * Q_b) / (2.0 * h);
	dQ_dn *= (SIDE == BoundarySide::Lower) ? -1.0 : 1.0;
	Vec t1{}, t2{};
	detail::transverseDerivatives<problem_t, DIR>(consVar, b0, geom, t1, t2); // at the boundary cell of this ghost cell's row
	const amrex::Real L = geom.prob_domain.length(d);
	Vec dQ = detail::fromNormalFrame<problem_t, DIR>(detail::outflowNormalDerivative<problem_t, SIDE>(
	    detail::toNormalFrame<problem_t, DIR>(Q_b), detail::toNormalFrame<problem_t, DIR>(dQ_dn), detail::toNormalFrame<problem_t, DIR>(t1),
	    detail::toNormalFrame<problem_t, DIR>(t2), P_outflow, L));
	dQ *= (SIDE == BoundarySide::Lower) ? -1.0 : 1.0; // towards the ghost cells
	const int steps = (here[d] - edge[d]) * (-inward);
	Vec ghost{};
	if (steps >= 1 && steps <= 4) {
		ghost = detail::continuation<problem_t>(Q_1, Q_b, h * dQ, steps);
	}
	detail::storeGhost<problem_t>(consVar, i, j, k, ghost);
}

// low-order variant: far-field pressure imposed on a copy of the boundary cell; if the gas flows INTO the domain the face reflects instead
template <typename problem_t, FluxDir DIR, BoundarySide SIDE>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE void setOutflowBoundaryLowOrder(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar,
								    amrex::GeometryData const &geom, const amrex::Real P_outflow)
{
	using Vec = detail::PrimVec<problem_t>;
	constexpr int d = static_cast<int>(DIR);
	constexpr int inward = (SIDE == BoundarySide::Lower) ? 1 : -1;
	auto [i, j, k] = iv.dim3();
	const int here[3] = {i, j, k};
	const auto edge = (SIDE == BoundarySide::Lower) ? geom.Domain().loVect3d() : geom.Domain().hiVect3d();
	const int steps = (here[d] - edge[d]) * (-inward);
	int b0[3] = {i, j, k};
	b0[d] = edge[d];
	Vec Q_b = detail::primAt<problem_t>(consVar, b0);
	const amrex::Real v_normal = detail::toNormalFrame<problem_t, DIR>(Q_b)[1];
	Vec ghost{};
	if (((SIDE == BoundarySide::Lower) && (v_normal > 0.)) || ((SIDE == BoundarySide::Upper) && (v_normal < 0.))) {
		// mirror image: ghost cell number s is interior cell number s - 1 with the normal velocity reversed
		if (steps >= 1 && steps <= 4) {
			int m[3] = {i, j, k};
			m[d] = edge[d] + (steps - 1) * inward;
			Vec q = detail::toNormalFrame<problem_t, DIR>(detail::primAt<problem_t>(consVar, m));
			q[1] *= -1.0;
			ghost = detail::fromNormalFrame<problem_t, DIR>(q);
		}
	} else if (steps >= 1 && steps <= 4) {
		Q_b[4] = P_outflow;
		ghost = Q_b;
	}
	detail::storeGhost<problem_t>(consVar, i, j, k, ghost);
}

// subsonic inflow through the lower x face relaxing to (T_t, u_t, v_t, w_t, s_t); ideal gas with constant k_B / mu
template <typename problem_t>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE void setInflowX1Lower(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar, amrex::GeometryData const &geom,
							  const amrex::Real T_t, const amrex::Real u_t, const amrex::Real v_t, const amrex::Real w_t,
							  amrex::GpuArray<amrex::Real, HydroSystem<problem_t>::nscalars_> const &s_t)
{
	using Vec = detail::PrimVec<problem_t>;
	auto [i, j, k] = iv.dim3();
	const int ilo = geom.Domain().loVect3d()[0];
	const amrex::Real h = geom.CellSize(0);
	const amrex::Real L = geom.prob_domain.length(0);
	const int c0[3] = {ilo, j, k}, c1[3] = {ilo + 1, j, k}, c2[3] = {ilo + 2, j, k};
	const Vec Q_b = detail::primAt<problem_t>(consVar, c0);
	const Vec Q_1 = detail::primAt<problem_t>(consVar, c1);
	const Vec Q_2 = detail::primAt<problem_t>(consVar, c2);
	const Vec dQ_data = (-3. * Q_b + 4. * Q_1 - Q_2) / (2. * h);

	const amrex::Real rho = Q_b[0], u = Q_b[1], v = Q_b[2], w = Q_b[3], P = Q_b[4], Eaux = Q_b[5];
	const amrex::Real T = quokka::EOS<problem_t>::ComputeTgasFromEint(rho, quokka::EOS<problem_t>::ComputeEintFromPres(rho, P));
	const amrex::Real Eaux_t = quokka::EOS<problem_t>::ComputeEintFromTgas(rho, T_t);
	const amrex::Real c = quokka::EOS<problem_t>::ComputeSoundSpeed(rho, P);
	const amrex::Real M = std::clamp(std::sqrt(u * u + v * v + w * w) / c, 0., 1.);
	const amrex::Real eta = 2.; // every relaxation coefficient of the reference is 2
	const amrex::Real R = quokka::EOS_Traits<problem_t>::boltzmann_constant / quokka::EOS_Traits<problem_t>::mean_molecular_weight;
	const amrex::Real rc = rho * c;
	// leaving: the u - c wave, from the interior; entering: the u + c wave relaxes u to u_t
	const amrex::Real Aminus = dQ_data[4] - rc * dQ_data[1];
	const amrex::Real Aplus = (u != 0.) ? -(c * c) * eta * rho * (M * M - 1) * (u - u_t) / (L * (c + u)) : (c * c) * eta * rho * u_t * (M * M - 1) / (L * c);
	Vec d{};
	d[4] = 0.5 * (Aminus + Aplus);
	d[1] = 0.5 * (Aplus - Aminus) / rc;
	if (u != 0.) {
		const amrex::Real S = -eta * R * rc * (T - T_t) / (L * u); // entropy wave: temperature relaxes to T_t
		d[0] = (S + 0.5 * (Aminus + Aplus)) / (c * c);
		d[2] = c * eta * (v - v_t) / (L * u);
		d[3] = c * eta * (w - w_t) / (L * u);
		d[5] = c * eta * (Eaux - Eaux_t) / (L * u);
		for (int n = 0; n < HydroSystem<problem_t>::nscalars_; ++n) {
			d[6 + n] = c * eta * (Q_b[6 + n] - s_t[n]) / (L * u);
		}
	} else {
		d[0] = 0.5 * (Aminus + Aplus) / (c * c);
	}
	const int steps = ilo - i;
	Vec ghost{};
	if (steps >= 1 && steps <= 4) {
		ghost = detail::continuation<problem_t>(Q_1, Q_b, -1.0 * (h * d), steps);
	}
	detail::storeGhost<problem_t>(consVar, i, j, k, ghost);
}

// low-order variant: density extrapolated, velocity, temperature and scalars prescribed
template <typename problem_t>
AMREX_GPU_DEVICE AMREX_FORCE_INLINE void setInflowX1LowerLowOrder(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar,
								  amrex::GeometryData const &geom, const amrex::Real T_t, const amrex::Real u_t, const amrex::Real v_t,
								  const amrex::Real w_t, amrex::GpuArray<amrex::Real, HydroSystem<problem_t>::nscalars_> const &s_t)
{
	auto [i, j, k] = iv.dim3();
	const int c0[3] = {geom.Domain().loVect3d()[0], j, k};
	const auto Q_b = detail::primAt<problem_t>(consVar, c0);
	const amrex::Real rho = Q_b[0];
	const amrex::Real Eint = quokka::EOS<problem_t>::ComputeEintFromTgas(rho, T_t);
	detail::PrimVec<problem_t> ghost{};
	ghost[0] = rho;
	ghost[1] = u_t;
	ghost[2] = v_t;
	ghost[3] = w_t;
	ghost[4] = quokka::EOS<problem_t>::ComputePressure(rho, Eint);
	ghost[5] = Eint;
	for (int n = 0; n < HydroSystem<problem_t>::nscalars_; ++n) {
		ghost[6 + n] = s_t[n];
	}
	detail::storeGhost<problem_t>(consVar, i, j, k, ghost);
}
} // namespace NSCBC

#endif // QK_HOST_NSCBC_HPP_
