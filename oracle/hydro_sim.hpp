// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// hydro_sim.hpp: restatement of the hydro part of the level-0 driver
//   reference src/QuokkaSimulation.hpp:885-990   advanceHydroAtLevelWithRetries
//   reference src/QuokkaSimulation.hpp:992-1013  isCflViolated
//   reference src/QuokkaSimulation.hpp:1032-1322 advanceHydroAtLevel (RK2-SSP + FOFC)
//   reference src/QuokkaSimulation.hpp:1324-1368 replaceFluxes
//   reference src/QuokkaSimulation.hpp:1403-1490 computeHydroFluxes
//   reference src/QuokkaSimulation.hpp:1492-1517 hydroFluxFunction
//   reference src/QuokkaSimulation.hpp:1519-1568 computeFOHydroFluxes / hydroFOFluxFunction
//   reference src/QuokkaSimulation.hpp:408-441   computeMaxSignalLocal (hydro-only branch)
//   reference src/simulation.hpp:703-818         computeTimestepAtLevel / computeTimestep
//   reference src/simulation.hpp:827-981         evolve (time loop + figure of merit)
// Uniform grid only (max_level = 0): no flux registers, no subcycling, no regrid.
#ifndef ORACLE_HYDRO_SIM_HPP_
#define ORACLE_HYDRO_SIM_HPP_

#include <functional>
#include <array>
#include <chrono>
#include <cstdlib>
#include <map>
#include <optional>
#include <string>
#include <cmath>
#include <cstdio>
#include <limits>
#include <utility>
#include <vector>

#include "boundary.hpp"
#include "grid.hpp"
#include "hydro.hpp"
#include "hydro_fused.hpp"
#include "radiation.hpp"
#include "radiation_multigroup.hpp"

namespace oracle
{

using FluxArrays = std::array<MultiFab, 3>;

// (box, sub-range) tasks for OpenMP: the range of every box cut into slabs along its slowest active dimension.  Every
// operator below writes each output cell / face from exactly one loop index, so disjoint sub-ranges never race, and the
// arithmetic per cell is unchanged.  (cpu_baseline leg of bench.py: 64 boxes alone cannot occupy a 256-core host.)
inline auto slabTasks(int nboxes, std::function<Box(int)> const &rangeOf, int ndim, int slab = 4) -> std::vector<std::pair<int, Box>>
{
	std::vector<std::pair<int, Box>> tasks;
	const int d = ndim - 1;
	for (int b = 0; b < nboxes; ++b) {
		Box const r = rangeOf(b);
		for (int lo = r.lo[d]; lo <= r.hi[d]; lo += slab) {
			Box piece = r;
			piece.lo[d] = lo;
			piece.hi[d] = std::min(lo + slab - 1, r.hi[d]);
			tasks.emplace_back(b, piece);
		}
	}
	return tasks;
}

struct HydroSim {
	// --- configuration (public data members of AMRSimulation / QuokkaSimulation) ---
	HydroSystem hydro;
	Geometry geom;
	std::vector<Box> grids;
	std::vector<BCRec> BCs_cc;
	CustomBCFunc customBC;

	int ncomp_cc = kNumHydroVars;		 // Physics_Indices::nvarTotal_cc
	int nghost_cc = 4;			 // simulation.hpp:363
	double stopTime_ = 1.0;			 // simulation.hpp:153
	double cflNumber_ = 0.3;		 // simulation.hpp:154
	long maxTimesteps_ = 10000;		 // simulation.hpp:157
	double maxDt_ = std::numeric_limits<double>::max();
	double initDt_ = std::numeric_limits<double>::max();
	double constantDt_ = 0.0;
	double densityFloor_ = 0.0;
	double tempFloor_ = 0.0;
	int integratorOrder_ = 2;		 // QuokkaSimulation.hpp:136
	int reconstructionOrder_ = 3;		 // :137
	int useDualEnergy_ = 1;			 // :139
	int abortOnFofcFailure_ = 1;		 // :140
	double artificialViscosityK_ = 0.;	 // :141
	int verbose = 0;

	// --- radiation (QuokkaSimulation.hpp:125-133; Physics_Traits::is_radiation_enabled) ---
	bool is_radiation_enabled = false;
	bool is_hydro_enabled = true; // Physics_Traits::is_hydro_enabled (false: radiation-only problems, e.g. RadStreaming)
	RadSystem rad;
	double radiationCflNumber_ = 0.3;
	int maxSubsteps_ = 10;
	int radiationReconstructionOrder_ = 3;
	bool use_wavespeed_correction_ = false; // QuokkaSimulation.hpp:133
	// RadSystem<problem_t>::SetRadEnergySource(radEnergySource, indexRange, dx, prob_lo, prob_hi, time): user hook
	std::function<void(Array4<double> const &, Box const &, Geometry const &, double)> SetRadEnergySource;
	long radiationCellUpdates_ = 0;
	long rad_iteration_counter[4] = {0, 0, 0, 0};	      // solves, Newton iterations, max Newton iterations, (decoupled)
	long rad_iteration_failure_counter[3] = {0, 0, 0}; // coupling, dust, outer

	// --- linear advection (reference src/linear_advection/AdvectionSimulation.hpp:89-96): one scalar carried at (advectionVx_, Vy_, Vz_); the
	// hydro / radiation members above are unused in this mode
	bool is_mhd_enabled = false; // Physics_Traits::is_mhd_enabled: selects HLLD in hydroFluxFunction (QuokkaSimulation.hpp:1512)
	bool is_advection = false;
	double advectionV[3] = {1.0, 0.0, 0.0};

	// --- state ---
	MultiFab state_old_cc_;
	MultiFab state_new_cc_;
	double tNew_ = 0.0;
	double dt_ = 1.e100; // simulation.hpp:449
	long istep = 0;
	long cellUpdates_ = 0;
	// diagnostics (counters the tests look at)
	long fofc1_cells = 0, fofc2_cells = 0, retries = 0;

	[[nodiscard]] auto ndim() const -> int { return geom.ndim; }
	[[nodiscard]] auto ncompHydro() const -> int { return hydro.tr.nvar(); }

	void define()
	{
		g_spacedim = ndim();
		state_old_cc_ = MultiFab(grids, ncomp_cc, nghost_cc, ndim());
		state_new_cc_ = MultiFab(grids, ncomp_cc, nghost_cc, ndim());
	}

	[[nodiscard]] auto CountCells() const -> long
	{
		long n = 0;
		for (auto const &b : grids) {
			n += b.numPts();
		}
		return n;
	}

	void fillBC(MultiFab &state, double time) { fillBoundaryConditions(state, geom, BCs_cc, customBC, time); }

	// simulation.hpp:1608-1626 (after the user's setInitialConditionsOnGrid has filled state_new_cc_)
	void finishInitialConditions()
	{
		fillBC(state_new_cc_, 0.0);
		state_old_cc_ = state_new_cc_;
	}

	// QuokkaSimulation.hpp:1492-1517
	void hydroFluxFunction(int dir, MultiFab const &primVar, MultiFab &leftState, MultiFab &rightState, MultiFab &flux, MultiFab &faceVel,
			       MultiFab const &x1Flat, MultiFab const &x2Flat, MultiFab const &x3Flat, int ng_reconstruct, int nvars) const
	{
		auto const cells = slabTasks(primVar.size(), [&](int b) { return grow(primVar.valid[b], ng_reconstruct, ndim()); }, ndim());
		_Pragma("omp parallel for schedule(dynamic)")
		for (size_t t = 0; t < cells.size(); ++t) {
			int const b = cells[t].first;
			Box const &cellRange = cells[t].second;
			if (reconstructionOrder_ == 3) {
				ReconstructStatesPPM(dir, primVar.const_array(b), leftState.array(b), rightState.array(b), cellRange, nvars);
			} else if (reconstructionOrder_ == 2) {
				ReconstructStatesPLM(dir, lim_minmod, primVar.const_array(b), leftState.array(b), rightState.array(b), cellRange, nvars);
			} else {
				ReconstructStatesConstant(dir, primVar.const_array(b), leftState.array(b), rightState.array(b), cellRange, nvars);
			}
		}
		_Pragma("omp parallel for schedule(dynamic)")
		for (size_t t = 0; t < cells.size(); ++t) {
			int const b = cells[t].first;
			hydro.FlattenShocks(dir, primVar.const_array(b), x1Flat.const_array(b), x2Flat.const_array(b), x3Flat.const_array(b),
					    leftState.array(b), rightState.array(b), cells[t].second, nvars);
		}
		auto const faces = slabTasks(primVar.size(), [&](int b) { return flux.validbox(b); }, ndim());
		_Pragma("omp parallel for schedule(dynamic)")
		for (size_t t = 0; t < faces.size(); ++t) {
			int const b = faces[t].first;
			// QuokkaSimulation.hpp:1512-1516: HLLD for MHD problems (the reference's stub: B = 0), HLLC otherwise
			hydro.ComputeFluxes(is_mhd_enabled ? riemann_HLLD : riemann_HLLC, dir, flux.array(b), faceVel.array(b), leftState.const_array(b), rightState.const_array(b),
					    primVar.const_array(b), artificialViscosityK_, faces[t].second);
		}
	}

	// QuokkaSimulation.hpp:1403-1490
	// the fused, vectorised form of computeHydroFluxes (hydro_fused.hpp; bench.py's cpu_baseline leg): same fluxes in every bit, one pass per box
	bool use_fused_fluxes = false;
	bool use_fused_stages = true; // with use_fused_fluxes: the whole stage box by box (fusedStage) where it applies
	auto computeHydroFluxesFused(MultiFab const &consVar, int nvars) const -> std::pair<FluxArrays, FluxArrays>
	{
		auto fv = takeFluxSet(nvars); // every face is overwritten below: nothing to clear
		FluxArrays &flux = fv.first;
		FluxArrays &facevel = fv.second;
		const int nb = consVar.size();
		_Pragma("omp parallel")
		{
			fused::Work work; // (one set of box-sized scratch arrays per thread)
			_Pragma("omp for schedule(dynamic)")
			for (int b = 0; b < nb; ++b) {
				fusedHydroFluxesBox(hydro.tr, consVar.const_array(b), grids[b], {flux[0].array(b), flux[1].array(b), flux[2].array(b)},
						    {facevel[0].array(b), facevel[1].array(b), facevel[2].array(b)}, artificialViscosityK_, work);
			}
		}
		return fv;
	}

	// One RK stage of the fused leg, box by box with no pass over the level in between: the stage's fluxes of `fluxInput` (stage 1: kept in `fv` and
	// flux_rk2 = 0.5 F; stage 2: flux_rk2 += 0.5 F), then the update stateNew = limits(stateOld + dt rhs) from `fv` (stage 1) or from the
	// accumulated flux_rk2 / avgFaceVel (stage 2).  Returns the number of cells PredictStep flagged (then the caller takes the operator path).
	auto fusedStage(int stage, MultiFab const &fluxInput, MultiFab const &stateOld, MultiFab &stateNew, std::pair<FluxArrays, FluxArrays> &fv, FluxArrays &flux_rk2,
			FluxArrays &avgFaceVel, double dt_lev, iMultiFab &redoFlag) const -> long
	{
		const int nb = fluxInput.size();
		long nbad = 0;
		_Pragma("omp parallel")
		{
			fused::Work work; // (one set of box-sized scratch arrays per thread)
			_Pragma("omp for schedule(dynamic) reduction(+ : nbad)")
			for (int b = 0; b < nb; ++b) {
				std::array<Array4<double>, 3> F{}, V{};
				if (stage == 1) {
					F = {fv.first[0].array(b), fv.first[1].array(b), fv.first[2].array(b)};
					V = {fv.second[0].array(b), fv.second[1].array(b), fv.second[2].array(b)};
				}
				std::array<Array4<double>, 3> const Fa{flux_rk2[0].array(b), flux_rk2[1].array(b), flux_rk2[2].array(b)};
				std::array<Array4<double>, 3> const Va{avgFaceVel[0].array(b), avgFaceVel[1].array(b), avgFaceVel[2].array(b)};
				fusedHydroFluxesBox(hydro.tr, fluxInput.const_array(b), grids[b], F, V, artificialViscosityK_, work, stage, Fa, Va);
				FluxArrays const &uf = (stage == 1) ? fv.first : flux_rk2;
				FluxArrays const &uv = (stage == 1) ? fv.second : avgFaceVel;
				nbad += fusedHydroUpdateBox(hydro.tr, stateOld.const_array(b), stateNew.array(b), grids[b],
							    {uf[0].const_array(b), uf[1].const_array(b), uf[2].const_array(b)},
							    {uv[0].const_array(b), uv[1].const_array(b), uv[2].const_array(b)}, geom.dx, dt_lev, densityFloor_, tempFloor_,
							    useDualEnergy_ == 1, redoFlag.array(b));
			}
		}
		return nbad;
	}

	// a (flux, face velocity) set whose every face the caller overwrites: from the pool when one is there
	auto takeFluxSet(int nvars) const -> std::pair<FluxArrays, FluxArrays>
	{
		FluxArrays flux, facevel;
		if (!pool_.fluxes.empty() && pool_.fluxes.back().first[0].nc == nvars) {
			flux = std::move(pool_.fluxes.back().first);
			facevel = std::move(pool_.fluxes.back().second);
			pool_.fluxes.pop_back();
		} else {
			for (int idim = 0; idim < 3; ++idim) {
				flux[idim] = MultiFab(grids, nvars, 0, ndim(), idim);
				facevel[idim] = MultiFab(grids, 1, 0, ndim(), idim);
			}
		}
		return std::make_pair(std::move(flux), std::move(facevel));
	}

	auto computeHydroFluxes(MultiFab const &consVar, int nvars) const -> std::pair<FluxArrays, FluxArrays>
	{
		if (use_fused_fluxes && consVar.ng == 4 && fusedHydroFluxesApplicable(hydro.tr, reconstructionOrder_, is_mhd_enabled)) {
			return computeHydroFluxesFused(consVar, nvars);
		}
		const int flatteningGhost = 2;
		const int reconstructGhost = 1;
		MultiFab primVar(grids, nvars, nghost_cc, ndim());
		std::array<MultiFab, 3> flatCoefs;
		FluxArrays flux, facevel, leftState, rightState;
		for (int idim = 0; idim < 3; ++idim) {
			flatCoefs[idim] = MultiFab(grids, 1, flatteningGhost, ndim());
		}
		for (int idim = 0; idim < ndim(); ++idim) {
			leftState[idim] = MultiFab(grids, nvars, reconstructGhost, ndim(), idim);
			rightState[idim] = MultiFab(grids, nvars, reconstructGhost, ndim(), idim);
			flux[idim] = MultiFab(grids, nvars, 0, ndim(), idim);
			facevel[idim] = MultiFab(grids, 1, 0, ndim(), idim);
		}
		auto const all = slabTasks(consVar.size(), [&](int b) { return grow(grids[b], nghost_cc, ndim()); }, ndim());
		_Pragma("omp parallel for schedule(dynamic)")
		for (size_t t = 0; t < all.size(); ++t) {
			hydro.ConservedToPrimitive(consVar.const_array(all[t].first), primVar.array(all[t].first), all[t].second);
		}
		auto const flat = slabTasks(consVar.size(), [&](int b) { return grow(grids[b], flatteningGhost, ndim()); }, ndim());
		for (int idim = 0; idim < ndim(); ++idim) {
			_Pragma("omp parallel for schedule(dynamic)")
			for (size_t t = 0; t < flat.size(); ++t) {
				hydro.ComputeFlatteningCoefficients(idim, primVar.const_array(flat[t].first), flatCoefs[idim].array(flat[t].first), flat[t].second);
			}
		}
		for (int idim = 0; idim < ndim(); ++idim) {
			hydroFluxFunction(idim, primVar, leftState[idim], rightState[idim], flux[idim], facevel[idim], flatCoefs[0], flatCoefs[1], flatCoefs[2],
					  reconstructGhost, nvars);
		}
		return std::make_pair(std::move(flux), std::move(facevel));
	}

	// QuokkaSimulation.hpp:1519-1568
	auto computeFOHydroFluxes(MultiFab const &consVar, int nvars) const -> std::pair<FluxArrays, FluxArrays>
	{
		const int reconstructRange = 1;
		MultiFab primVar(grids, nvars, nghost_cc, ndim());
		FluxArrays flux, facevel, leftState, rightState;
		for (int idim = 0; idim < ndim(); ++idim) {
			leftState[idim] = MultiFab(grids, nvars, reconstructRange, ndim(), idim);
			rightState[idim] = MultiFab(grids, nvars, reconstructRange, ndim(), idim);
			flux[idim] = MultiFab(grids, nvars, 0, ndim(), idim);
			facevel[idim] = MultiFab(grids, 1, 0, ndim(), idim);
		}
		auto const all = slabTasks(consVar.size(), [&](int b) { return grow(grids[b], nghost_cc, ndim()); }, ndim());
		_Pragma("omp parallel for schedule(dynamic)")
		for (size_t t = 0; t < all.size(); ++t) {
			hydro.ConservedToPrimitive(consVar.const_array(all[t].first), primVar.array(all[t].first), all[t].second);
		}
		auto const cells = slabTasks(consVar.size(), [&](int b) { return grow(grids[b], reconstructRange, ndim()); }, ndim());
		for (int idim = 0; idim < ndim(); ++idim) {
			_Pragma("omp parallel for schedule(dynamic)")
			for (size_t t = 0; t < cells.size(); ++t) {
				int const b = cells[t].first;
				ReconstructStatesConstant(idim, primVar.const_array(b), leftState[idim].array(b), rightState[idim].array(b), cells[t].second, nvars);
			}
			auto const faces = slabTasks(consVar.size(), [&](int b) { return flux[idim].validbox(b); }, ndim());
			_Pragma("omp parallel for schedule(dynamic)")
			for (size_t t = 0; t < faces.size(); ++t) {
				int const b = faces[t].first;
				hydro.ComputeFluxes(riemann_LLF, idim, flux[idim].array(b), facevel[idim].array(b), leftState[idim].const_array(b),
						    rightState[idim].const_array(b), primVar.const_array(b), artificialViscosityK_, faces[t].second);
			}
		}
		return std::make_pair(std::move(flux), std::move(facevel));
	}

	// MultiFab::Saxpy(dst, a, src, 0, 0, ncomp, 0): dst += a*src on valid (face) boxes
	static void Saxpy(MultiFab &dst, double a, MultiFab const &src, int ncomp)
	{
		_Pragma("omp parallel for schedule(dynamic)")
		for (int b = 0; b < dst.size(); ++b) {
			auto d = dst.array(b);
			auto s = src.const_array(b);
			Box const r = dst.validbox(b);
			for (int n = 0; n < ncomp; ++n) {
				for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
					for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
						for (int i = r.lo[0]; i <= r.hi[0]; ++i) {
							d(i, j, k, n) += a * s(i, j, k, n);
						}
					}
				}
			}
		}
	}

	// QuokkaSimulation.hpp:1324-1368 (redoFlag has 1 ghost cell, filled by FillBoundary)
	void replaceFluxes(FluxArrays &fluxes, FluxArrays const &FOfluxes, iMultiFab const &redoFlag) const
	{
		for (int idim = 0; idim < ndim(); ++idim) {
			int const ncomp = fluxes[idim].nc;
			for (int b = 0; b < redoFlag.size(); ++b) {
				auto flux_arr = fluxes[idim].array(b);
				auto FO_arr = FOfluxes[idim].const_array(b);
				auto flag = redoFlag.const_array(b);
				Box const r = grow(redoFlag.valid[b], 1, ndim());
				for (int n = 0; n < ncomp; ++n) {
					for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
						for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
							for (int i = r.lo[0]; i <= r.hi[0]; ++i) {
								if (flag(i, j, k) == redo_redo) {
									if (flux_arr.contains(i, j, k)) {
										flux_arr(i, j, k, n) = FO_arr(i, j, k, n);
									}
									int const ip = i + (idim == 0 ? 1 : 0);
									int const jp = j + (idim == 1 ? 1 : 0);
									int const kp = k + (idim == 2 ? 1 : 0);
									if (flux_arr.contains(ip, jp, kp)) {
										flux_arr(ip, jp, kp, n) = FO_arr(ip, jp, kp, n);
									}
								}
							}
						}
					}
				}
			}
		}
	}

	static auto sumFlags(iMultiFab const &redoFlag) -> long
	{
		long s = 0;
		for (int b = 0; b < redoFlag.size(); ++b) {
			auto f = redoFlag.const_array(b);
			Box const &r = redoFlag.valid[b];
			for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
				for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
					for (int i = r.lo[0]; i <= r.hi[0]; ++i) {
						s += f(i, j, k);
					}
				}
			}
		}
		return s;
	}

	void rhsPdvPredict(MultiFab &rhs, FluxArrays const &fluxes, FluxArrays const &faceVel, MultiFab const &stateOld, MultiFab &stateNew, double dt_lev,
			   iMultiFab &redoFlag) const
	{
		auto const tasks = slabTasks(rhs.size(), [&](int b) { return grids[b]; }, ndim());
		_Pragma("omp parallel for schedule(dynamic)")
		for (size_t t = 0; t < tasks.size(); ++t) {
			int const b = tasks[t].first;
			std::array<Array4<const double>, 3> f{}, v{};
			for (int d = 0; d < ndim(); ++d) {
				f[d] = fluxes[d].const_array(b);
				v[d] = faceVel[d].const_array(b);
			}
			Box const &r = tasks[t].second;
			hydro.ComputeRhsFromFluxes(rhs.array(b), f, geom.dx, ncompHydro(), r);
			hydro.AddInternalEnergyPdV(rhs.array(b), stateOld.const_array(b), geom.dx, v, redoFlag.const_array(b), r);
			hydro.PredictStep(stateOld.const_array(b), stateNew.array(b), rhs.const_array(b), dt_lev, ncompHydro(), redoFlag.array(b), r);
		}
	}

	void limitsAndSync(MultiFab &state) const
	{
		auto const tasks = slabTasks(state.size(), [&](int b) { return grids[b]; }, ndim());
		_Pragma("omp parallel for schedule(dynamic)")
		for (size_t t = 0; t < tasks.size(); ++t) {
			hydro.EnforceLimits(densityFloor_, tempFloor_, state.array(tasks[t].first), tasks[t].second);
		}
		if (useDualEnergy_ == 1) {
			_Pragma("omp parallel for schedule(dynamic)")
			for (size_t t = 0; t < tasks.size(); ++t) {
				hydro.SyncDualEnergy(state.array(tasks[t].first), tasks[t].second);
			}
		}
	}

	// QuokkaSimulation.hpp:992-1013
	auto isCflViolated(double dt_actual) const -> bool
	{
		double max_signal = -std::numeric_limits<double>::infinity();
		_Pragma("omp parallel for schedule(dynamic) reduction(max : max_signal)")
		for (int b = 0; b < state_new_cc_.size(); ++b) {
			max_signal = std::max(max_signal, hydro.maxSignalSpeedLocal(state_new_cc_.const_array(b), grids[b]));
		}
		const double dx_min = minDx();
		const double dt_cfl = cflNumber_ * (dx_min / max_signal);
		const double max_factor = 1.1;
		return dt_actual > (max_factor * dt_cfl);
	}

	[[nodiscard]] auto minDx() const -> double
	{
		double m = geom.dx[0];
		for (int d = 1; d < ndim(); ++d) {
			m = std::min(m, geom.dx[d]);
		}
		return m;
	}

	// QuokkaSimulation.hpp:1032-1322 (no Strang sources, no flux registers, no tracers)
	// wall time per phase of advanceHydroAtLevel (ORACLE_PROF=1; printed by the destructor): where a CPU step spends its time
	mutable std::map<std::string, double> profSeconds_, profMin_;
	mutable std::map<std::string, long> profCalls_;
	bool const prof_ = std::getenv("ORACLE_PROF") != nullptr;
	struct ProfScope {
		HydroSim const *s;
		const char *name;
		std::chrono::steady_clock::time_point t0;
		ProfScope(HydroSim const *sim, const char *n) : s(sim), name(n), t0(std::chrono::steady_clock::now()) {}
		~ProfScope()
		{
			if (s->prof_) {
				double const el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
				s->profSeconds_[name] += el;
				auto it = s->profMin_.find(name);
				if (it == s->profMin_.end() || el < it->second) {
					s->profMin_[name] = el;
				}
				++s->profCalls_[name];
			}
		}
	};
	~HydroSim()
	{
		if (prof_) {
			for (auto const &kv : profSeconds_) {
				std::fprintf(stderr, "[oracle prof] %-28s %9.3f s in %5ld calls, fastest call %8.3f ms\n", kv.first.c_str(), kv.second, profCalls_[kv.first],
					     1e3 * profMin_[kv.first]);
			}
		}
	}

	// The fused leg keeps the temporaries of a step between steps (the reference takes them from AMReX's pooled arena; plain std::vector
	// allocations of this size come back from the kernel as fresh zero pages, whose first touch cost more than the arithmetic of the step):
	// same sizes, cleared where the step accumulates into them — the same values as fresh arrays.
	struct StepPool {
		std::optional<MultiFab> inter, rhs, oldTmp;
		std::optional<iMultiFab> redo;
		FluxArrays flux_rk2, avgVel;
		std::vector<std::pair<FluxArrays, FluxArrays>> fluxes; // (flux, face velocity) sets handed out by computeHydroFluxesFused and given back
		bool defined = false;
	};
	mutable StepPool pool_;

	auto advanceHydroAtLevel(MultiFab &state_old_cc_tmp, double time, double dt_lev) -> bool
	{
		const int nc = ncompHydro();
		std::optional<ProfScope> psAlloc;
		psAlloc.emplace(this, "allocate + clear");
		const bool pooled = use_fused_fluxes;
		std::optional<MultiFab> localInter;
		if (pooled && !pool_.inter) {
			pool_.inter.emplace(grids, ncomp_cc, nghost_cc, ndim());
		}
		if (!pooled) {
			localInter.emplace(grids, ncomp_cc, nghost_cc, ndim());
		}
		MultiFab &state_inter_cc_ = pooled ? *pool_.inter : *localInter;
		// whole stages box by box (fusedStage) where the fused flux evaluation applies and the state is the six hydro variables
		const bool stagesFused = pooled && use_fused_stages && state_old_cc_tmp.ng == 4 && ncomp_cc == nc && nc == 6 &&
					 fusedHydroFluxesApplicable(hydro.tr, reconstructionOrder_, is_mhd_enabled);
		if (!(stagesFused && pool_.defined)) { // (a fused stage writes every valid cell and the ghost fill the same ghost cells every step: what is
						       //  never written stays the zero of the first step's clear)
			state_inter_cc_.setVal(0);
		}

		FluxArrays localRk2, localVel;
		FluxArrays &flux_rk2 = pooled ? pool_.flux_rk2 : localRk2;
		FluxArrays &avgFaceVel = pooled ? pool_.avgVel : localVel;
		// The reference allocates avgFaceVel with 2 ghost faces "for tracer particles"
		// (QuokkaSimulation.hpp:1061); the ghosts only feed tracers (out of scope) and make
		// replaceFluxes read FOfaceVel out of bounds (:1352 `contains` on the ghosted array), so
		// the oracle allocates none.  Valid faces are unaffected.
		const int nghost_vel = 0;
		for (int idim = 0; idim < ndim(); ++idim) {
			if (!(pooled && pool_.defined)) {
				flux_rk2[idim] = MultiFab(grids, nc, 0, ndim(), idim);
				avgFaceVel[idim] = MultiFab(grids, 1, nghost_vel, ndim(), idim);
			}
			if (!stagesFused) { // (stage 1 of the fused leg SETS them to 0.0 + 0.5 F)
				flux_rk2[idim].setVal(0);
				avgFaceVel[idim].setVal(0);
			}
		}
		if (pooled && !pool_.defined) {
			pool_.rhs.emplace(grids, nc, 0, ndim());
			pool_.redo.emplace(grids, 1, 1, ndim());
			pool_.defined = true;
		}
		psAlloc.reset();

		// :1076 update ghost zones [old timestep]
		{
			ProfScope const ps(this, "fill ghosts");
			fillBC(state_old_cc_tmp, time);
		}

		// :1096 — the first-order fluxes of the old state.  The reference evaluates them before stage 1 whether or not a stage needs them; they
		// depend on the old state alone, which no stage changes, so the fused leg (use_fused_fluxes) evaluates them when a stage first flags
		// cells — the same values, in the deck's 1000 steps never needed
		std::optional<std::pair<FluxArrays, FluxArrays>> FO;
		auto needFO = [&]() -> std::pair<FluxArrays, FluxArrays> & {
			if (!FO) {
				ProfScope const ps(this, "first-order fluxes");
				FO.emplace(computeFOHydroFluxes(state_old_cc_tmp, nc));
			}
			return *FO;
		};
		if (!use_fused_fluxes) {
			needFO();
		}

		// Stage 1 of RK2-SSP (:1099-1198)
		{
			auto const &stateOld = state_old_cc_tmp;
			auto &stateNew = state_inter_cc_;
			auto fv1 = [&] {
				if (stagesFused) {
					return takeFluxSet(nc);
				}
				ProfScope const ps(this, "hydro fluxes");
				return computeHydroFluxes(stateOld, nc);
			}();
			auto &fluxArrays = fv1.first;
			auto &faceVel = fv1.second;

			if (!stagesFused) {
				ProfScope const ps(this, "flux_rk2 += 0.5 F");
				for (int idim = 0; idim < ndim(); ++idim) {
					Saxpy(flux_rk2[idim], 0.5, fluxArrays[idim], nc);
					Saxpy(avgFaceVel[idim], 0.5, faceVel[idim], 1);
				}
			}

			std::optional<MultiFab> localRhs;
			std::optional<iMultiFab> localRedo;
			if (!pooled) {
				localRhs.emplace(grids, nc, 0, ndim());
				localRedo.emplace(grids, 1, 1, ndim());
			}
			MultiFab &rhs = pooled ? *pool_.rhs : *localRhs;
			iMultiFab &redoFlag = pooled ? *pool_.redo : *localRedo;
			redoFlag.setVal(redo_none);

			long ncells_bad = 0;
			if (stagesFused) {
				ProfScope const ps(this, "fused stage");
				ncells_bad = fusedStage(1, stateOld, stateOld, stateNew, fv1, flux_rk2, avgFaceVel, dt_lev, redoFlag);
			} else {
				ProfScope const ps(this, "rhs + PdV + PredictStep");
				rhsPdvPredict(rhs, fluxArrays, faceVel, stateOld, stateNew, dt_lev, redoFlag);
				ncells_bad = sumFlags(redoFlag);
			}
			if (ncells_bad > 0) {
				fofc1_cells += ncells_bad;
				if (verbose != 0) {
					std::printf("[FOFC-1] flux correcting %ld cells\n", ncells_bad);
				}
				FillBoundary(redoFlag, geom);
				replaceFluxes(fluxArrays, needFO().first, redoFlag);
				replaceFluxes(faceVel, needFO().second, redoFlag);
				rhsPdvPredict(rhs, fluxArrays, faceVel, stateOld, stateNew, dt_lev, redoFlag);
				long const ncells_bad2 = sumFlags(redoFlag);
				if (ncells_bad2 > 0) {
					if (abortOnFofcFailure_ != 0) {
						return false;
					}
				}
			}
			if (!stagesFused || ncells_bad > 0) { // (a fused stage applied them itself; after a flux correction the operator path redid the cells)
				ProfScope const ps(this, "limits + dual energy");
				limitsAndSync(stateNew);
			}
			if (pooled) {
				pool_.fluxes.push_back(std::move(fv1)); // (handed out again by the next flux evaluation)
			}
		}

		// Stage 2 of RK2-SSP (:1202-1287)
		if (integratorOrder_ == 2) {
			{
				ProfScope const ps(this, "fill ghosts");
				fillBC(state_inter_cc_, time + dt_lev);
			}

			auto const &stateOld = state_old_cc_tmp;
			auto const &stateInter = state_inter_cc_;
			auto &stateFinal = state_new_cc_;
			auto fv2 = [&]() -> std::pair<FluxArrays, FluxArrays> {
				if (stagesFused) {
					return {}; // (stage 2 of the fused leg adds 0.5 F to flux_rk2 as it forms F and keeps no F)
				}
				ProfScope const ps(this, "hydro fluxes");
				return computeHydroFluxes(stateInter, nc);
			}();
			auto &fluxArrays = fv2.first;
			auto &faceVel = fv2.second;

			if (!stagesFused) {
				ProfScope const ps(this, "flux_rk2 += 0.5 F");
				for (int idim = 0; idim < ndim(); ++idim) {
					Saxpy(flux_rk2[idim], 0.5, fluxArrays[idim], nc);
					Saxpy(avgFaceVel[idim], 0.5, faceVel[idim], 1);
				}
			}

			std::optional<MultiFab> localRhs;
			std::optional<iMultiFab> localRedo;
			if (!pooled) {
				localRhs.emplace(grids, nc, 0, ndim());
				localRedo.emplace(grids, 1, 1, ndim());
			}
			MultiFab &rhs = pooled ? *pool_.rhs : *localRhs;
			iMultiFab &redoFlag = pooled ? *pool_.redo : *localRedo;
			redoFlag.setVal(redo_none);

			long ncells_bad = 0;
			if (stagesFused) {
				ProfScope const ps(this, "fused stage");
				ncells_bad = fusedStage(2, stateInter, stateOld, stateFinal, fv2, flux_rk2, avgFaceVel, dt_lev, redoFlag);
			} else {
				ProfScope const ps(this, "rhs + PdV + PredictStep");
				rhsPdvPredict(rhs, flux_rk2, avgFaceVel, stateOld, stateFinal, dt_lev, redoFlag);
				ncells_bad = sumFlags(redoFlag);
			}
			if (ncells_bad > 0) {
				fofc2_cells += ncells_bad;
				if (verbose != 0) {
					std::printf("[FOFC-2] flux correcting %ld cells\n", ncells_bad);
				}
				FillBoundary(redoFlag, geom);
				replaceFluxes(flux_rk2, needFO().first, redoFlag);
				replaceFluxes(avgFaceVel, needFO().second, redoFlag);
				rhsPdvPredict(rhs, flux_rk2, avgFaceVel, stateOld, stateFinal, dt_lev, redoFlag);
				long const ncells_bad2 = sumFlags(redoFlag);
				if (ncells_bad2 > 0) {
					if (abortOnFofcFailure_ != 0) {
						return false;
					}
				}
			}
			if (!stagesFused || ncells_bad > 0) {
				ProfScope const ps(this, "limits + dual energy");
				limitsAndSync(stateFinal);
			}
			if (pooled && !stagesFused) {
				pool_.fluxes.push_back(std::move(fv2));
			}
		} else {
			// :1289 forward Euler: copy hydro comps of the valid region
			copyValid(state_new_cc_, state_inter_cc_, nc);
		}

		// :1321
		ProfScope const ps(this, "isCflViolated");
		return !isCflViolated(dt_lev);
	}

	void copyValid(MultiFab &dst, MultiFab const &src, int ncomp) const
	{
		_Pragma("omp parallel for schedule(dynamic)")
		for (int b = 0; b < dst.size(); ++b) {
			auto d = dst.array(b);
			auto s = src.const_array(b);
			Box const &r = grids[b];
			for (int n = 0; n < ncomp; ++n) {
				for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
					for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
						for (int i = r.lo[0]; i <= r.hi[0]; ++i) {
							d(i, j, k, n) = s(i, j, k, n);
						}
					}
				}
			}
		}
	}

	// QuokkaSimulation.hpp:885-990
	auto advanceHydroAtLevelWithRetries(double time, double dt_lev) -> bool
	{
		const int max_retries = 6;
		bool success = false;
		for (int retry_count = 0; retry_count <= max_retries; ++retry_count) {
			const int nsubsteps = static_cast<int>(std::pow(2, retry_count));
			const double dt_step = dt_lev / nsubsteps;
			if (retry_count > 0) {
				++retries;
			}
			// :939-940 temporary copy of the old state (with ghosts); the fused leg copies into an array it keeps (StepPool)
			std::optional<MultiFab> localTmp;
			std::optional<ProfScope> psCopy;
			psCopy.emplace(this, "copy of the old state");
			if (use_fused_fluxes) {
				if (!pool_.oldTmp) {
					pool_.oldTmp.emplace(state_old_cc_);
				} else {
					const int nb = state_old_cc_.size();
					_Pragma("omp parallel for schedule(static)")
					for (int b = 0; b < nb; ++b) {
						std::copy(state_old_cc_.fabs[b].d.begin(), state_old_cc_.fabs[b].d.end(), pool_.oldTmp->fabs[b].d.begin());
					}
				}
			} else {
				localTmp.emplace(state_old_cc_);
			}
			MultiFab &state_old_cc_tmp = use_fused_fluxes ? *pool_.oldTmp : *localTmp;
			psCopy.reset();
			for (int substep = 0; substep < nsubsteps; ++substep) {
				if (substep > 0) {
					// :947 amrex::Copy(state_old_cc_tmp, state_new_cc_, 0, 0, ncompHydro_, nghost_cc_)
					for (int b = 0; b < state_old_cc_tmp.size(); ++b) {
						auto d = state_old_cc_tmp.array(b);
						auto s = state_new_cc_.const_array(b);
						Box const r = state_old_cc_tmp.fabs[b].bx;
						for (int n = 0; n < ncompHydro(); ++n) {
							for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
								for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
									for (int i = r.lo[0]; i <= r.hi[0]; ++i) {
										d(i, j, k, n) = s(i, j, k, n);
									}
								}
							}
						}
					}
				}
				success = advanceHydroAtLevel(state_old_cc_tmp, time, dt_step);
				if (!success) {
					break;
				}
			}
			if (success) {
				break;
			}
		}
		return success;
	}

	// simulation.hpp:703-720 with QuokkaSimulation.hpp:408-441 (hydro-only branch) and norminf
	[[nodiscard]] auto computeTimestepAtLevel() const -> double
	{
		double domain_signal_max = 0.0; // norminf of a non-negative field
		if (is_advection) { // LinearAdvectionSystem::ComputeMaxSignalSpeed (linear_advection.hpp:47-60): the same value in every cell
			domain_signal_max = std::sqrt(advectionV[0] * advectionV[0] + advectionV[1] * advectionV[1] + advectionV[2] * advectionV[2]);
			return cflNumber_ * (minDx() / domain_signal_max);
		}
		_Pragma("omp parallel for schedule(dynamic) reduction(max : domain_signal_max)")
		for (int b = 0; b < state_new_cc_.size(); ++b) {
			Fab<double> maxSignal(grids[b], 1);
			if (!is_hydro_enabled) { // QuokkaSimulation.hpp:421-424, radiation only: RadSystem::ComputeMaxSignalSpeed = c_hat in every cell
				domain_signal_max = std::max(domain_signal_max, std::abs(rad.rt.c_hat));
				continue;
			}
			hydro.ComputeMaxSignalSpeed(state_new_cc_.const_array(b), maxSignal.array(), grids[b]);
			for (double v : maxSignal.d) {
				if (is_radiation_enabled) {
					// QuokkaSimulation.hpp:421-434: RadSystem::ComputeMaxSignalSpeed = c_hat (radiation_system.hpp:616-624)
					const double maxSignalRadiation = rad.rt.c_hat / static_cast<double>(maxSubsteps_);
					v = std::max(maxSignalRadiation, v);
				}
				domain_signal_max = std::max(domain_signal_max, std::abs(v));
			}
		}
		const double dx_min = minDx();
		return cflNumber_ * (dx_min / domain_signal_max);
	}

	// simulation.hpp:722-818 (single level)
	void computeTimestep()
	{
		double dt_tmp = computeTimestepAtLevel();
		constexpr double change_max = 1.1;
		dt_tmp = std::min(dt_tmp, change_max * dt_);

		double dt_0 = dt_tmp;
		dt_0 = std::min(dt_0, 1.0 * dt_tmp); // n_factor * dt_tmp[0], n_factor = 1
		dt_0 = std::min(dt_0, maxDt_);
		if (tNew_ == 0.0) {
			dt_0 = std::min(dt_0, initDt_);
		}
		if (constantDt_ > 0.0) {
			dt_0 = constantDt_;
		}
		const double eps = 1.e-3 * dt_0;
		if (tNew_ + dt_0 > stopTime_ - eps) {
			dt_0 = stopTime_ - tNew_;
		}
		dt_ = dt_0;
	}


	// ------------------------------------------------------------------ radiation (QuokkaSimulation.hpp:397-406, 1570-1961)
	[[nodiscard]] auto computeNumberOfRadiationSubsteps(double dt_lev_hydro) const -> int
	{
		double const c_hat = rad.rt.c_hat;
		double const dx_min = minDx();
		double const dtrad_tmp = radiationCflNumber_ * (dx_min / c_hat);
		int const nsubSteps = static_cast<int>(std::ceil(dt_lev_hydro / dtrad_tmp));
		return nsubSteps;
	}

	// fluxFunction<DIR> + computeRadiationFluxes for one box (QuokkaSimulation.hpp:1884-1961)
	struct RadFluxes {
		std::array<Fab<double>, 3> flux, fluxDiffusive;
	};
	[[nodiscard]] auto computeRadiationFluxes(Array4<const double> const &consVar, Box const &indexRange) const -> RadFluxes
	{
		RadFluxes out;
		const int nvars = rad.nRadComps(); // Physics_NumVars::numRadVars * nGroups
		for (int dir = 0; dir < ndim(); ++dir) {
			Box const ghostRange = grow(indexRange, nghost_cc, ndim());
			Box const reconstructRange = grow(indexRange, 1, ndim());
			Box const x1ReconstructRange = faceBox(reconstructRange, dir);
			Fab<double> primVar(ghostRange, nvars);
			Fab<double> x1LeftState(x1ReconstructRange, nvars);
			Fab<double> x1RightState(x1ReconstructRange, nvars);
			rad.ConservedToPrimitive(consVar, primVar.array(), ghostRange);
			if (radiationReconstructionOrder_ == 3) {
				ReconstructStatesPPM(dir, primVar.const_array(), x1LeftState.array(), x1RightState.array(), reconstructRange, nvars);
			} else if (radiationReconstructionOrder_ == 2) {
				ReconstructStatesPLM(dir, lim_MC, primVar.const_array(), x1LeftState.array(), x1RightState.array(), x1ReconstructRange, nvars);
			} else {
				ReconstructStatesConstant(dir, primVar.const_array(), x1LeftState.array(), x1RightState.array(), x1ReconstructRange, nvars);
			}
			Box const x1FluxRange = faceBox(indexRange, dir);
			out.flux[dir] = Fab<double>(x1FluxRange, nvars);
			out.fluxDiffusive[dir] = Fab<double>(x1FluxRange, nvars);
			rad.ComputeFluxes(dir, out.flux[dir].array(), out.fluxDiffusive[dir].array(), x1LeftState.const_array(), x1RightState.const_array(),
					  x1FluxRange, consVar, geom.dx, use_wavespeed_correction_); // (:1958-1960)
		}
		return out;
	}

	// QuokkaSimulation.hpp:1790-1821
	void advanceRadiationForwardEuler(double time, double dt_radiation)
	{
		fillBC(state_old_cc_, time);
		_Pragma("omp parallel for schedule(dynamic)")
		for (int b = 0; b < state_new_cc_.size(); ++b) {
			auto fl = computeRadiationFluxes(state_old_cc_.const_array(b), grids[b]);
			std::array<Array4<const double>, 3> f{};
			for (int d = 0; d < ndim(); ++d) {
				f[d] = fl.flux[d].const_array();
			}
			rad.PredictStep(state_old_cc_.const_array(b), state_new_cc_.array(b), f, dt_radiation, geom.dx, grids[b]);
		}
	}

	// QuokkaSimulation.hpp:1823-1857
	void advanceRadiationMidpointRK2(double time, double dt_radiation)
	{
		fillBC(state_new_cc_, time + dt_radiation);
		_Pragma("omp parallel for schedule(dynamic)")
		for (int b = 0; b < state_new_cc_.size(); ++b) {
			auto flOld = computeRadiationFluxes(state_old_cc_.const_array(b), grids[b]);
			auto fl = computeRadiationFluxes(state_new_cc_.const_array(b), grids[b]);
			std::array<Array4<const double>, 3> f0{}, f1{};
			for (int d = 0; d < ndim(); ++d) {
				f0[d] = flOld.flux[d].const_array();
				f1[d] = fl.flux[d].const_array();
			}
			rad.AddFluxesRK2(state_new_cc_.array(b), state_old_cc_.const_array(b), state_new_cc_.const_array(b), f0, f1, dt_radiation, geom.dx,
					 grids[b]);
		}
	}

	// QuokkaSimulation.hpp:1859-1882
	void operatorSplitSourceTerms(double time, double dt, int stage)
	{
		for (int b = 0; b < state_new_cc_.size(); ++b) {
			Fab<double> radEnergySource(grids[b], rad.nGroups_(), 0.0); // :1866-1869 (nGroups components)
			if (SetRadEnergySource) {
				SetRadEnergySource(radEnergySource.array(), grids[b], geom, time + dt);
			}
			int counter[4] = {0, 0, 0, 0};
			int failure[3] = {0, 0, 0};
			if (rad.nGroups_() <= 1) { // :1875-1881
				rad.AddSourceTermsSingleGroup(state_new_cc_.array(b), radEnergySource.const_array(), grids[b], dt, stage, counter, failure);
			} else {
				mg::MG(rad).AddSourceTermsMultiGroup(state_new_cc_.array(b), radEnergySource.const_array(), grids[b], dt, stage, counter, failure);
			}
			rad_iteration_counter[0] += counter[0];
			rad_iteration_counter[1] += counter[1];
			rad_iteration_counter[2] = std::max<long>(rad_iteration_counter[2], counter[2]);
			rad_iteration_counter[3] += counter[3];
			for (int n = 0; n < 3; ++n) {
				rad_iteration_failure_counter[n] += failure[n];
			}
		}
	}

	// QuokkaSimulation.hpp:1576-1722
	auto subcycleRadiationAtLevel(double time, double dt_lev_hydro) -> bool
	{
		int nsubSteps = 0;
		double dt_radiation = NAN;
		if (is_hydro_enabled && !(constantDt_ > 0.)) { // :1583 — radiation-only problems take one radiation step of dt_lev
			nsubSteps = computeNumberOfRadiationSubsteps(dt_lev_hydro);
			dt_radiation = dt_lev_hydro / static_cast<double>(nsubSteps);
		} else {
			dt_radiation = dt_lev_hydro;
			nsubSteps = 1;
		}
		if (!(nsubSteps >= 1 && nsubSteps <= (maxSubsteps_ + 1) && dt_radiation > 0.0)) {
			std::fprintf(stderr, "radiation substep assertion failed: nsubSteps = %d\n", nsubSteps);
			return false;
		}
		double time_subcycle = time;
		for (int i = 0; i < nsubSteps; ++i) {
			if (i > 0) {
				// swapRadiationState (:1570-1574): copy radiation comps of the valid region new -> old
				for (int b = 0; b < state_old_cc_.size(); ++b) {
					auto d = state_old_cc_.array(b);
					auto s = state_new_cc_.const_array(b);
					Box const &r = grids[b];
					for (int n = rad.nstartHyperbolic_; n < rad.nstartHyperbolic_ + rad.nRadComps(); ++n) {
						for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
							for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
								for (int ii = r.lo[0]; ii <= r.hi[0]; ++ii) {
									d(ii, j, k, n) = s(ii, j, k, n);
								}
							}
						}
					}
				}
			}
			advanceRadiationForwardEuler(time_subcycle, dt_radiation);
			if (IMEX_a22 > 0.0) {
				operatorSplitSourceTerms(time_subcycle, dt_radiation, 1);
			}
			advanceRadiationMidpointRK2(time_subcycle, dt_radiation);
			operatorSplitSourceTerms(time_subcycle, dt_radiation, 2);
			if (rad_iteration_failure_counter[0] > 0 || rad_iteration_failure_counter[1] > 0 || rad_iteration_failure_counter[2] > 0) {
				std::fprintf(stderr, "Newton-Raphson / outer iteration for matter-radiation coupling failed to converge!\n");
				return false;
			}
			time_subcycle += dt_radiation;
			radiationCellUpdates_ += CountCells();
		}
		return true;
	}

	// one coarse step: simulation.hpp:866-890 + :1276-1286 + QuokkaSimulation.hpp:653-707
	// ------------------------------------------------------------------ linear advection
	// AdvectionSimulation::computeFluxes + fluxFunction<DIR> (AdvectionSimulation.hpp:385-434): ConservedToPrimitive is the identity
	// (linear_advection.hpp:62-70), PPM reconstruction, upwind flux (linear_advection.hpp:165-198).  One Fab per box and direction.
	[[nodiscard]] auto advectionFluxes(MultiFab const &consVar) const -> std::vector<std::array<Fab<double>, 3>>
	{
		std::vector<std::array<Fab<double>, 3>> out(static_cast<size_t>(consVar.size()));
		const int nvars = ncomp_cc;
		const int reconstructRange = 1;
		for (int b = 0; b < consVar.size(); ++b) {
			Box const &valid = grids[b];
			for (int dir = 0; dir < ndim(); ++dir) {
				Box const cellRange = grow(valid, reconstructRange, ndim());
				Box const faceRange = faceBox(cellRange, dir);
				Fab<double> left(faceRange, nvars), right(faceRange, nvars);
				ReconstructStatesPPM(dir, consVar.const_array(b), left.array(), right.array(), cellRange, nvars);
				out[static_cast<size_t>(b)][static_cast<size_t>(dir)] = Fab<double>(faceBox(valid, dir), nvars);
				auto F = out[static_cast<size_t>(b)][static_cast<size_t>(dir)].array();
				auto L = left.const_array();
				auto R = right.const_array();
				double const vx = advectionV[dir];
				Box const fb = faceBox(valid, dir);
				for (int n = 0; n < nvars; ++n) {
					for (int k = fb.lo[2]; k <= fb.hi[2]; ++k) {
						for (int j = fb.lo[1]; j <= fb.hi[1]; ++j) {
							for (int i = fb.lo[0]; i <= fb.hi[0]; ++i) {
								// upwind side of the interface (the array element of a face is the same in every view)
								F(i, j, k, n) = (vx < 0.0) ? vx * R(i, j, k, n) : vx * L(i, j, k, n);
							}
						}
					}
				}
			}
		}
		return out;
	}

	// AdvectionSimulation::advanceSingleTimestepAtLevel (AdvectionSimulation.hpp:236-383, single level, integratorOrder_ = 2; the swap of
	// old and new state happened in step())
	void advanceAdvectionAtLevel(double time, double dt_lev)
	{
		const int nvars = ncomp_cc;
		double const dtdx[3] = {dt_lev / geom.dx[0], dt_lev / geom.dx[1], dt_lev / geom.dx[2]};
		auto diff = [&](Array4<const double> const &F, int dir, int i, int j, int k, int n) {
			return F(i, j, k, n) - F(i + (dir == 0 ? 1 : 0), j + (dir == 1 ? 1 : 0), k + (dir == 2 ? 1 : 0), n);
		};
		fillBC(state_old_cc_, time);
		{ // LinearAdvectionSystem::PredictStep (linear_advection.hpp:82-118)
			auto const flux = advectionFluxes(state_old_cc_);
			for (int b = 0; b < state_new_cc_.size(); ++b) {
				auto Uo = state_old_cc_.const_array(b);
				auto Un = state_new_cc_.array(b);
				Box const &r = grids[b];
				for (int n = 0; n < nvars; ++n) {
					for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
						for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
							for (int i = r.lo[0]; i <= r.hi[0]; ++i) {
								double s = dtdx[0] * diff(flux[static_cast<size_t>(b)][0].const_array(), 0, i, j, k, n);
								if (ndim() >= 2) {
									s = s + dtdx[1] * diff(flux[static_cast<size_t>(b)][1].const_array(), 1, i, j, k, n);
								}
								if (ndim() == 3) {
									s = s + dtdx[2] * diff(flux[static_cast<size_t>(b)][2].const_array(), 2, i, j, k, n);
								}
								Un(i, j, k, n) = Uo(i, j, k, n) + s;
							}
						}
					}
				}
			}
		}
		fillBC(state_new_cc_, time + dt_lev);
		{ // LinearAdvectionSystem::AddFluxesRK2 (linear_advection.hpp:120-163): U_new = (0.5 U_0 + 0.5 U_1) + (0.5 FxU_1 + 0.5 FyU_1 + 0.5 FzU_1)
			auto const flux = advectionFluxes(state_new_cc_);
			for (int b = 0; b < state_new_cc_.size(); ++b) {
				auto U0 = state_old_cc_.const_array(b);
				auto Un = state_new_cc_.array(b);
				Box const &r = grids[b];
				for (int n = 0; n < nvars; ++n) {
					for (int k = r.lo[2]; k <= r.hi[2]; ++k) {
						for (int j = r.lo[1]; j <= r.hi[1]; ++j) {
							for (int i = r.lo[0]; i <= r.hi[0]; ++i) {
								double const U_0 = U0(i, j, k, n);
								double const U_1 = Un(i, j, k, n);
								double s = 0.5 * (dtdx[0] * diff(flux[static_cast<size_t>(b)][0].const_array(), 0, i, j, k, n));
								if (ndim() >= 2) {
									s = s + 0.5 * (dtdx[1] * diff(flux[static_cast<size_t>(b)][1].const_array(), 1, i, j, k, n));
								}
								if (ndim() == 3) {
									s = s + 0.5 * (dtdx[2] * diff(flux[static_cast<size_t>(b)][2].const_array(), 2, i, j, k, n));
								}
								Un(i, j, k, n) = (0.5 * U_0 + 0.5 * U_1) + s;
							}
						}
					}
				}
			}
		}
	}

	auto step() -> bool
	{
		g_spacedim = ndim(); // (the reference's AMREX_SPACEDIM: a 2-D build permutes the X2 views differently, hyperbolic.hpp)
		ProfScope const psStep(this, "WHOLE STEP");
		{
			ProfScope const ps(this, "computeTimestep");
			computeTimestep();
		}
		double const time = tNew_;
		tNew_ += dt_;
		std::swap(state_old_cc_, state_new_cc_);
		bool ok = true;
		if (is_advection) {
			advanceAdvectionAtLevel(time, dt_);
			++istep;
			cellUpdates_ += CountCells();
			return true;
		}
		if (is_hydro_enabled) {
			ok = advanceHydroAtLevelWithRetries(time, dt_);
		} else { // QuokkaSimulation.hpp:681-685: copy hydro vars from state_old_cc_ to state_new_cc_
			for (int b = 0; b < state_new_cc_.size(); ++b) {
				auto const &src = state_old_cc_.fabs[b].d;
				auto &dst = state_new_cc_.fabs[b].d;
				size_t const n = src.size() / static_cast<size_t>(ncomp_cc) * kNumHydroVars;
				std::copy(src.begin(), src.begin() + static_cast<long>(n), dst.begin());
			}
		}
		if (ok && is_radiation_enabled) {
			ok = subcycleRadiationAtLevel(time, dt_); // QuokkaSimulation.hpp:689-692
		}
		++istep;
		cellUpdates_ += CountCells();
		return ok;
	}

	// simulation.hpp:856-951 time loop
	auto evolve() -> bool
	{
		double cur_time = tNew_;
		for (long s = istep; s < maxTimesteps_ && cur_time < stopTime_; ++s) {
			if (!step()) {
				return false;
			}
			cur_time += dt_;
			tNew_ = cur_time;
			if (cur_time >= stopTime_ - 1.e-6 * dt_) {
				break;
			}
		}
		return true;
	}
};

} // namespace oracle

#endif // ORACLE_HYDRO_SIM_HPP_
