// ORACLE — TEST INFRASTRUCTURE ONLY (see grid.hpp header).
//
// oracle_capi.cpp: extern "C" entry points so tests/ (ctypes + numpy) can drive the CPU
// restatement.  Arrays are Fortran-order, component outermost (amrex::Array4 layout), FP64.
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <memory>

#include "amr.hpp"
#include "cooling.hpp"
#include "hydro_sim.hpp"
#include "problems.hpp"

using namespace oracle;

extern "C" {

struct orc_hydro_traits {
	double gamma;
	double cs_isothermal;
	double mean_molecular_weight;
	double boltzmann_constant;
	int reconstruct_eint;
	int nscalars;
	int ndim;
};

static auto makeSystem(orc_hydro_traits const *t) -> HydroSystem
{
	HydroSystem h;
	h.tr.eos.tr.gamma = t->gamma;
	h.tr.eos.tr.cs_isothermal = t->cs_isothermal;
	h.tr.eos.tr.mean_molecular_weight = t->mean_molecular_weight;
	h.tr.eos.tr.boltzmann_constant = t->boltzmann_constant;
	h.tr.reconstruct_eint = (t->reconstruct_eint != 0);
	h.tr.nscalars = t->nscalars;
	h.tr.ndim = t->ndim;
	return h;
}

static auto mkbox(int const lo[3], int const hi[3]) -> Box
{
	Box b;
	for (int d = 0; d < 3; ++d) {
		b.lo[d] = lo[d];
		b.hi[d] = hi[d];
	}
	return b;
}

int orc_eos_variant() { return kEosVariant; }

// ------------------------------------------------------------------ per-operator entry points
// `cons`/`prim` live on `gbox` = valid box grown by nghost (in the first ndim dims).

void orc_cons_to_prim(orc_hydro_traits const *t, double const *cons, double *prim, int const glo[3], int const ghi[3])
{
	HydroSystem h = makeSystem(t);
	Box g = mkbox(glo, ghi);
	h.ConservedToPrimitive(Array4<const double>(cons, g, h.tr.nvar()), Array4<double>(prim, g, h.tr.nvar()), g);
}

void orc_flattening_coefficients(orc_hydro_traits const *t, int dir, double const *prim, int const glo[3], int const ghi[3], double *chi,
				 int const clo[3], int const chi_hi[3])
{
	g_spacedim = t->ndim;
	HydroSystem h = makeSystem(t);
	Box g = mkbox(glo, ghi);
	Box c = mkbox(clo, chi_hi);
	h.ComputeFlatteningCoefficients(dir, Array4<const double>(prim, g, h.tr.nvar()), Array4<double>(chi, c, 1), c);
}

// Full flux evaluation for one box exactly as computeHydroFluxes / computeFOHydroFluxes do it.
// order: 1/2/3 -> donor/PLM-minmod/PPM with flattening + HLLC ; order = 0 -> first-order LLF path
// (computeFOHydroFluxes).  flux_d: face box (nodal in d) x nvar ; fvel_d: face box x 1.
void orc_compute_hydro_fluxes(orc_hydro_traits const *t, int order, double K_visc, double const *cons, int const vlo[3], int const vhi[3], int nghost,
			      double *flux0, double *flux1, double *flux2, double *fvel0, double *fvel1, double *fvel2)
{
	g_spacedim = t->ndim;
	HydroSim sim;
	sim.hydro = makeSystem(t);
	sim.geom.ndim = t->ndim;
	sim.grids = {mkbox(vlo, vhi)};
	sim.nghost_cc = nghost;
	sim.ncomp_cc = sim.hydro.tr.nvar();
	sim.artificialViscosityK_ = K_visc;
	sim.reconstructionOrder_ = (order == 0) ? 1 : (order % 10);
	sim.is_mhd_enabled = (order >= 10); // order + 10: the fluxes of an MHD problem (HLLD with the reference's B = 0 stub)
	order = order % 10;
	int const nv = sim.hydro.tr.nvar();
	MultiFab consMF(sim.grids, nv, nghost, t->ndim);
	std::memcpy(consMF.fabs[0].d.data(), cons, consMF.fabs[0].d.size() * sizeof(double));
	auto result = (order == 0) ? sim.computeFOHydroFluxes(consMF, nv) : sim.computeHydroFluxes(consMF, nv);
	double *fo[3] = {flux0, flux1, flux2};
	double *vo[3] = {fvel0, fvel1, fvel2};
	for (int d = 0; d < t->ndim; ++d) {
		std::memcpy(fo[d], result.first[d].fabs[0].d.data(), result.first[d].fabs[0].d.size() * sizeof(double));
		std::memcpy(vo[d], result.second[d].fabs[0].d.data(), result.second[d].fabs[0].d.size() * sizeof(double));
	}
}

// ------------------------------------------------------------------ multigroup helper functions (unit checks against independent formulas)
double orc_planck_integral(double x) { return planck::integrate_planck_from_0_to_x(x); }
double orc_planck_table_entry(int j) { return planck::Y_interp[j]; }

static auto mgSystem(int ngroups, double const *boundaries, double energy_unit, double k_B, double a_rad, double Erad_floor, int opacity_model) -> RadSystem
{
	RadSystem rs;
	rs.rt.nGroups = ngroups;
	rs.rt.radBoundaries.assign(boundaries, boundaries + ngroups + 1);
	rs.rt.energy_unit = energy_unit;
	rs.rt.radiation_constant = a_rad;
	rs.rt.Erad_floor = Erad_floor;
	rs.rt.opacity_model = opacity_model;
	rs.eos.tr.boltzmann_constant = k_B;
	return rs;
}
// ComputePlanckEnergyFractions / ComputeThermalRadiationMultiGroup (fractions, then E_g = max(a T^4 fraction_g, floor / nGroups))
void orc_planck_fractions(int ngroups, double const *boundaries, double energy_unit, double k_B, double a_rad, double Erad_floor, double T, double *fractions,
			  double *Erad_g)
{
	RadSystem const rs = mgSystem(ngroups, boundaries, energy_unit, k_B, a_rad, Erad_floor, 1);
	mg::MG const m(rs);
	auto const f = m.ComputePlanckEnergyFractions(m.boundaries(), T);
	auto const E = m.ComputeThermalRadiationMultiGroup(T, m.boundaries());
	for (int g = 0; g < ngroups; ++g) {
		fractions[g] = f[g];
		Erad_g[g] = E[g];
	}
}
double orc_planck_function(double energy_unit, double k_B, double a_rad, double nu, double T)
{
	double const b[2] = {1., 2.};
	RadSystem const rs = mgSystem(1, b, energy_unit, k_B, a_rad, 0., 1);
	return mg::MG(rs).PlanckFunction(nu, T);
}
void orc_group_mean_opacity(int ngroups, double const *boundaries, double const *expo, double const *lower, double const *alpha_quant, double *kappa)
{
	RadSystem const rs = mgSystem(ngroups, boundaries, 1., 1., 1., 0., 2);
	mg::MG const m(rs);
	mg::KappaExpoLower kel;
	kel.expo = mg::VA(ngroups + 1);
	kel.lower = mg::VA(ngroups + 1);
	mg::VA ratios(ngroups), alpha(ngroups);
	for (int g = 0; g < ngroups + 1; ++g) {
		kel.expo[g] = expo[g];
		kel.lower[g] = lower[g];
	}
	for (int g = 0; g < ngroups; ++g) {
		ratios[g] = boundaries[g + 1] / boundaries[g];
		alpha[g] = alpha_quant[g];
	}
	auto const k = m.ComputeGroupMeanOpacity(kel, ratios, alpha);
	for (int g = 0; g < ngroups; ++g) {
		kappa[g] = k[g];
	}
}
void orc_rad_quantity_exponents(int ngroups, double const *boundaries, double const *quant, double *exponents)
{
	RadSystem const rs = mgSystem(ngroups, boundaries, 1., 1., 1., 0., 3);
	mg::MG const m(rs);
	mg::VA q(ngroups);
	for (int g = 0; g < ngroups; ++g) {
		q[g] = quant[g];
	}
	auto const e = m.ComputeRadQuantityExponents(q, m.boundaries());
	for (int g = 0; g < ngroups; ++g) {
		exponents[g] = e[g];
	}
}

// ------------------------------------------------------------------ whole-simulation handle
struct orc_sim_config {
	int problem; // 0 = Sod, 1 = contact, 2 = Sedov
	int ndim;
	int n_cell[3];
	int max_grid_size[3];
	double prob_lo[3];
	double prob_hi[3];
	int periodic[3];
	// overrides (< 0 / NaN = keep the problem's default)
	double cfl;
	double stop_time;
	long max_timesteps;
	int reconstruction_order;
	int nscalars;
	// problem 3 (radiation-driven shell): the three columns of extern/dust_shell/initial_conditions.txt
	int table_len;
	double const *table_r, *table_Erad, *table_Frad;
	int rad_pow_mode; // RadTraits::pow_mode
	// problem 7 (parametrised 1-D hydro test): gamma, x_split, left[3], right[3], cfl, max_dt, init_dt, stop_time | profile, dirichlet
	double h1d[12];
	int h1d_i[2];
	double const *table_extra; // problem 17 (RadTube): fourth column of extern/pressure_tube/initial_conditions.txt (x, rho, Pgas, Erad)
	int opacity_model;	   // problems 16 / 18: OpacityModel override (<= 0: the problem file's choice)
};

// OpenMP team size for everything that follows (small problems run faster on one thread: every parallel region costs a barrier over
// the team, and a box that grants fewer CPUs than it shows makes an oversized team spin)
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
	omp_set_num_threads(n > 0 ? n : 1);
#else
	(void)n;
#endif
}

void *orc_sim_create(orc_sim_config const *c)
{
	auto sim = std::make_unique<HydroSim>();
	setupGeometry(*sim, c->ndim, c->n_cell, c->prob_lo, c->prob_hi, c->periodic, c->max_grid_size);
	if (c->problem == 0) {
		setupSod(*sim);
	} else if (c->problem == 1) {
		setupContact(*sim, c->nscalars > 0 ? c->nscalars : 0);
	} else if (c->problem == 2) {
		setupSedov(*sim);
	} else if (c->problem == 3) {
		setupShell(*sim, c->table_len, c->table_r, c->table_Erad, c->table_Frad);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 8) {
		setupMatterCoupling(*sim);
		if (c->h1d[0] > 0) { // RadMatterCouplingRSLA (test_radiation_matter_coupling_rsla.cpp:22,43): c_hat = 0.1 c, nothing else differs
			sim->rad.rt.c_hat = c->h1d[0] * sim->rad.rt.c_light;
		}
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 9) {
		setupSuOlson(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 7) {
		Hydro1DSpec p{};
		p.gamma = c->h1d[0];
		p.x_split = c->h1d[1];
		for (int n = 0; n < 3; ++n) {
			p.left[n] = c->h1d[2 + n];
			p.right[n] = c->h1d[5 + n];
		}
		p.cfl = c->h1d[8];
		p.max_dt = c->h1d[9];
		p.init_dt = c->h1d[10];
		p.stop_time = c->h1d[11];
		p.profile = c->h1d_i[0];
		p.dirichlet = c->h1d_i[1];
		p.max_timesteps = c->max_timesteps >= 0 ? c->max_timesteps : 100000;
		setupHydro1D(*sim, p);
	} else if (c->problem == 15) {
		setupShocktubeCMA(*sim);
	} else if (c->problem == 14) {
		setupRadPulse(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 13) {
		setupMarshakAsymptotic(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 12) {
		setupRadForce(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 11) {
		setupMarshak(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 10) {
		setupUniformAdvecting(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem >= 31 && c->problem <= 33) {
		setupAdvection(*sim, c->problem - 31);
	} else if (c->problem == 30) {
		setupGeneralOpacity(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 6) {
		setupScalarContact(*sim, c->nscalars > 0 ? c->nscalars : 1);
	} else if (c->problem == 5 || c->problem == 29) {
		setupStreaming(*sim, c->problem == 29 ? 1 : 0);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 4) {
		setupRadShock(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 16) {
		setupRadShockMG(*sim, c->opacity_model > 0 ? c->opacity_model : static_cast<int>(PPL_opacity_fixed_slope_spectrum));
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 17) {
		setupRadTube(*sim, c->table_len, c->table_r, c->table_Erad, c->table_Frad, c->table_extra);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 18) {
		setupMarshakVaytet(*sim, c->opacity_model > 0 ? c->opacity_model : static_cast<int>(PPL_opacity_full_spectrum));
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 25) {
		setupMarshakDust(*sim);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 23) {
		setupBlast2D(*sim);
	} else if (c->problem == 22) {
		setupQuirk(*sim);
	} else if (c->problem == 26 || c->problem == 27) { // h1d[1]: radiation.dust_gas_interaction_coeff of the deck
		setupLineCooling(*sim, c->problem == 27, c->h1d[1]);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 28) {
		setupMarshakDustPE(*sim, c->h1d[1]);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 21 || c->problem == 24) {
		setupRadDust(*sim, c->problem == 24);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else if (c->problem == 19 || c->problem == 20) {
		setupPulseMG(*sim, c->problem == 19);
		sim->rad.rt.pow_mode = c->rad_pow_mode;
	} else {
		return nullptr;
	}
	if ((c->problem == 4 || c->problem == 10) && c->h1d_i[0] > 0) {
		sim->rad.rt.beta_order = c->h1d_i[0]; // parity variants: the same problem with the O(beta^n) terms of another order
	}
	if (c->cfl > 0) {
		sim->cflNumber_ = c->cfl;
	}
	if (c->stop_time > 0) {
		sim->stopTime_ = c->stop_time;
	}
	if (c->max_timesteps >= 0) {
		sim->maxTimesteps_ = c->max_timesteps;
	}
	if (c->reconstruction_order > 0) {
		sim->reconstructionOrder_ = c->reconstruction_order;
	}
	return sim.release();
}

void orc_sim_destroy(void *p) { delete static_cast<HydroSim *>(p); }
int orc_sim_nboxes(void *p) { return static_cast<HydroSim *>(p)->state_new_cc_.size(); }
int orc_sim_ncomp(void *p) { return static_cast<HydroSim *>(p)->ncomp_cc; }
int orc_sim_nghost(void *p) { return static_cast<HydroSim *>(p)->nghost_cc; }
void orc_sim_box(void *p, int b, int lo[3], int hi[3])
{
	auto *s = static_cast<HydroSim *>(p);
	for (int d = 0; d < 3; ++d) {
		lo[d] = s->grids[b].lo[d];
		hi[d] = s->grids[b].hi[d];
	}
}
// copy the whole fab (valid + ghosts) of state_new_cc_ / state_old_cc_
void orc_sim_get_state(void *p, int which, int b, double *out)
{
	auto *s = static_cast<HydroSim *>(p);
	auto const &mf = (which == 0) ? s->state_new_cc_ : s->state_old_cc_;
	std::memcpy(out, mf.fabs[b].d.data(), mf.fabs[b].d.size() * sizeof(double));
}
void orc_sim_set_state(void *p, int which, int b, double const *in)
{
	auto *s = static_cast<HydroSim *>(p);
	auto &mf = (which == 0) ? s->state_new_cc_ : s->state_old_cc_;
	std::memcpy(mf.fabs[b].d.data(), in, mf.fabs[b].d.size() * sizeof(double));
}
void orc_sim_fill_ghosts(void *p, int which, double time)
{
	auto *s = static_cast<HydroSim *>(p);
	s->fillBC((which == 0) ? s->state_new_cc_ : s->state_old_cc_, time);
}
void orc_sim_set_rad_reconstruction_order(void *p, int order) { static_cast<HydroSim *>(p)->radiationReconstructionOrder_ = order; }
void orc_sim_set_wavespeed_correction(void *p, int on) { static_cast<HydroSim *>(p)->use_wavespeed_correction_ = (on != 0); }
double orc_sim_time(void *p) { return static_cast<HydroSim *>(p)->tNew_; }
double orc_sim_dt(void *p) { return static_cast<HydroSim *>(p)->dt_; }
long orc_sim_istep(void *p) { return static_cast<HydroSim *>(p)->istep; }
long orc_sim_cell_updates(void *p) { return static_cast<HydroSim *>(p)->cellUpdates_; }
double orc_sim_compute_dt(void *p)
{
	auto *s = static_cast<HydroSim *>(p);
	double const save = s->dt_;
	s->computeTimestep();
	double const r = s->dt_;
	s->dt_ = save;
	return r;
}
void orc_sim_counters(void *p, long out[3])
{
	auto *s = static_cast<HydroSim *>(p);
	out[0] = s->fofc1_cells;
	out[1] = s->fofc2_cells;
	out[2] = s->retries;
}
// the two transport stages of one radiation substep without the source terms (advanceRadiationForwardEuler + advanceRadiationMidpointRK2,
// QuokkaSimulation.hpp:1790-1857), state_old = state_new on entry as after swapRadiationState
void orc_sim_rad_transport_only(void *p, double dt_radiation)
{
	auto *s = static_cast<HydroSim *>(p);
	g_spacedim = s->ndim();
	s->advanceRadiationForwardEuler(s->tNew_, dt_radiation);
	s->advanceRadiationMidpointRK2(s->tNew_, dt_radiation);
}
// the fused, vectorised flux evaluation (hydro_fused.hpp) instead of the operator sequence: same bits (tests/test_oracle_fused_cpu.py)
// the floors of EnforceLimits (QuokkaSimulation.hpp densityFloor_, tempFloor_: 0 by default)
void orc_sim_set_limits(void *p, double densityFloor, double tempFloor)
{
	static_cast<HydroSim *>(p)->densityFloor_ = densityFloor;
	static_cast<HydroSim *>(p)->tempFloor_ = tempFloor;
}
// on: 0 operator form; 1 fused flux evaluation only; 3 whole stages box by box as well (HydroSim::fusedStage)
void orc_sim_set_fused_fluxes(void *p, int on)
{
	static_cast<HydroSim *>(p)->use_fused_fluxes = ((on & 1) != 0);
	static_cast<HydroSim *>(p)->use_fused_stages = ((on & 2) != 0);
}
// both forms of the flux evaluation on the sim's CURRENT state_new (ghost cells filled here): flux[d] as [6][faces] + face velocity [faces], x fastest
int orc_sim_hydro_fluxes(void *p, int fused, int b, int dir, double *flux_out, double *vel_out)
{
	auto *s = static_cast<HydroSim *>(p);
	s->fillBC(s->state_new_cc_, s->tNew_);
	bool const keep = s->use_fused_fluxes;
	s->use_fused_fluxes = (fused != 0);
	auto fv = s->computeHydroFluxes(s->state_new_cc_, s->ncompHydro());
	s->use_fused_fluxes = keep;
	auto const &F = fv.first[dir].fabs[b].d;
	auto const &V = fv.second[dir].fabs[b].d;
	std::copy(F.begin(), F.end(), flux_out);
	std::copy(V.begin(), V.end(), vel_out);
	return static_cast<int>(V.size());
}
// one coarse step with the reference's own dt control; returns 1 on success
int orc_sim_step(void *p)
{
	auto *s = static_cast<HydroSim *>(p);
	bool const ok = s->step();
	return ok ? 1 : 0;
}
// nsteps coarse steps, recording after each one the time and all components of the valid cell (i, j, k) of box b
// (what computeAfterTimestep of RadMatterCoupling collects); returns the number of steps taken
long orc_sim_run_record(void *p, long nsteps, int b, int i, int j, int k, double *out_t, double *out_u)
{
	auto *s = static_cast<HydroSim *>(p);
	long n = 0;
	double cur_time = s->tNew_;
	for (; n < nsteps && cur_time < s->stopTime_; ++n) {
		if (!s->step()) {
			break;
		}
		cur_time += s->dt_;
		s->tNew_ = cur_time;
		auto const a = s->state_new_cc_.const_array(b);
		out_t[n] = s->tNew_;
		for (int c = 0; c < s->ncomp_cc; ++c) {
			out_u[n * s->ncomp_cc + c] = a(i, j, k, c);
		}
		if (cur_time >= s->stopTime_ - 1.e-6 * s->dt_) {
			++n;
			break;
		}
	}
	return n;
}

// one hydro advance with a caller-supplied dt (no dt control): old <- new, advance
int orc_sim_advance_fixed_dt(void *p, double dt)
{
	auto *s = static_cast<HydroSim *>(p);
	g_spacedim = s->ndim();
	double const time = s->tNew_;
	s->dt_ = dt;
	s->tNew_ += dt;
	std::swap(s->state_old_cc_, s->state_new_cc_);
	bool const ok = s->advanceHydroAtLevelWithRetries(time, dt);
	++s->istep;
	s->cellUpdates_ += s->CountCells();
	return ok ? 1 : 0;
}
void orc_sim_rad_counters(void *p, long out[8])
{
	auto *s = static_cast<HydroSim *>(p);
	for (int n = 0; n < 4; ++n) {
		out[n] = s->rad_iteration_counter[n];
	}
	for (int n = 0; n < 3; ++n) {
		out[4 + n] = s->rad_iteration_failure_counter[n];
	}
	out[7] = s->radiationCellUpdates_;
}
// fills `out` (valid box of box b, 1 comp) with SetRadEnergySource at `time`
void orc_sim_rad_source(void *p, int b, double time, double *out)
{
	auto *s = static_cast<HydroSim *>(p);
	Array4<double> a(out, s->grids[b], 1);
	for (int64_t n = 0; n < s->grids[b].numPts(); ++n) {
		out[n] = 0.0;
	}
	if (s->SetRadEnergySource) {
		s->SetRadEnergySource(a, s->grids[b], s->geom, time);
	}
}
int orc_sim_evolve(void *p) { return static_cast<HydroSim *>(p)->evolve() ? 1 : 0; }

// ErrorEst of the gradient-threshold family on the (ghost-filled) new state of box b; `tags` covers the valid box, 1 char per cell
void orc_sim_tag_relative_gradient(void *p, int b, int field, double eta_threshold, double q_min, int min_inclusive, char *tags)
{
	auto *s = static_cast<HydroSim *>(p);
	Array4<char> t(tags, s->grids[b], 1);
	tagRelativeGradient(s->hydro, s->state_new_cc_.const_array(b), t, s->grids[b], s->ndim(), field, eta_threshold, q_min, min_inclusive != 0);
}

void orc_sim_tag_centered_gradient(void *p, int b, int comp, int dir, double dx, double eta_threshold, double q_min, int min_inclusive, char *tags)
{
	auto *s = static_cast<HydroSim *>(p);
	Array4<char> t(tags, s->grids[b], 1);
	tagCenteredGradient(s->state_new_cc_.const_array(b), t, s->grids[b], comp, dir, dx, eta_threshold, q_min, min_inclusive != 0);
}

// coarse -> fine interpolation of `region` (fine indices); arrays given with their lower / upper corners
void orc_interp_from_coarse(double *fine, const int *flo, const int *fhi, const double *crse_old, const double *crse_new, const int *clo, const int *chi,
			    int ncomp_total, const int *rlo, const int *rhi, double w_old, double w_new, int ncomp, int method, int hooks, int ndim, const int *ratio)
{
	Box fb, cb, region;
	for (int d = 0; d < 3; ++d) {
		fb.lo[d] = flo[d];
		fb.hi[d] = fhi[d];
		cb.lo[d] = clo[d];
		cb.hi[d] = chi[d];
		region.lo[d] = rlo[d];
		region.hi[d] = rhi[d];
	}
	Array4<double> f(fine, fb, ncomp_total);
	Array4<const double> co(crse_old, cb, ncomp_total), cn(crse_new, cb, ncomp_total);
	interpFromCoarse(f, co, cn, region, w_old, w_new, ncomp, method, hooks != 0, ndim, ratio);
}

// amrex::average_down of one fine array onto one coarse array; boxes as lo[3], hi[3] (with ghosts = the arrays' extents); region in coarse indices
void orc_average_down(const double *fine, const int *flo, const int *fhi, double *crse, const int *clo, const int *chi, int ncomp_total, const int *rlo,
		      const int *rhi, int scomp, int ncomp, const int *ratio)
{
	Box fb, cb, region;
	for (int d = 0; d < 3; ++d) {
		fb.lo[d] = flo[d];
		fb.hi[d] = fhi[d];
		cb.lo[d] = clo[d];
		cb.hi[d] = chi[d];
		region.lo[d] = rlo[d];
		region.hi[d] = rhi[d];
	}
	Array4<const double> f(fine, fb, ncomp_total);
	Array4<double> c(crse, cb, ncomp_total);
	averageDown(f, c, region, scomp, ncomp, ratio);
}

// ------------------------------------------------------------------ tabulated cooling (cooling.hpp)
// the five datasets of a cloudy_cooling_tools file as H5Dread delivers them (Parameter1[n0], Temperature[n1], Cooling / Heating / MMW [n0][n1])
void *orc_cloudy_create(int n0, int n1, const double *parameter1, const double *temperature, const double *cooling_rate, const double *heating_rate, const double *mmw)
{
	return new cooling::cloudy_tables(cooling::prepare_tables(n0, n1, parameter1, temperature, cooling_rate, heating_rate, mmw));
}
void orc_cloudy_destroy(void *p) { delete static_cast<cooling::cloudy_tables *>(p); }
// out: T_min, T_max, mmw_min, mmw_max
void orc_cloudy_ranges(void *p, double *out)
{
	auto const &t = *static_cast<cooling::cloudy_tables *>(p);
	out[0] = t.T_min;
	out[1] = t.T_max;
	out[2] = t.mmw_min;
	out[3] = t.mmw_max;
}
// the prepared arrays: which = 0 log_nH, 1 log_Tgas, 2 cooling, 3 heating, 4 mean molecular weight (n_H fastest)
void orc_cloudy_get(void *p, int which, double *out)
{
	auto const &t = *static_cast<cooling::cloudy_tables *>(p);
	std::vector<double> const *v[5] = {&t.log_nH_v, &t.log_Tgas_v, &t.cool_v, &t.heat_v, &t.mmw_v};
	std::copy(v[which]->begin(), v[which]->end(), out);
}
// what: 0 ComputeTgasFromEgas, 1 ComputeEgasFromTgas, 2 ComputeMMW, 3 ComputeCoolingLength, 4 cloudy_cooling_function (val = E_int, T, E_int, E_int, T)
void orc_cloudy_evaluate(void *p, double gamma, int what, long n, const double *rho, const double *val, double *out)
{
	auto const &t = *static_cast<cooling::cloudy_tables *>(p);
	_Pragma("omp parallel for schedule(dynamic, 64)")
	for (long i = 0; i < n; ++i) {
		switch (what) {
		case 0:
			out[i] = cooling::ComputeTgasFromEgas(rho[i], val[i], gamma, t);
			break;
		case 1:
			out[i] = cooling::ComputeEgasFromTgas(rho[i], val[i], gamma, t);
			break;
		case 2:
			out[i] = cooling::ComputeMMW(rho[i], val[i], gamma, t);
			break;
		case 3:
			out[i] = cooling::ComputeCoolingLength(rho[i], val[i], gamma, t);
			break;
		default:
			out[i] = cooling::cloudy_cooling_function(rho[i], val[i], t);
			break;
		}
	}
}
// computeCooling over n cells: U[6][n] = (rho, x1Mom, x2Mom, x3Mom, Egas, Eint_aux), component outermost; nsteps[n]
void orc_cloudy_compute_cooling(void *p, double gamma, double dt, double T_floor, long n, double *U, int *nsteps)
{
	auto const &t = *static_cast<cooling::cloudy_tables *>(p);
	_Pragma("omp parallel for schedule(dynamic, 16)")
	for (long i = 0; i < n; ++i) {
		double cell[6];
		for (int c = 0; c < 6; ++c) {
			cell[c] = U[c * n + i];
		}
		nsteps[i] = cooling::computeCoolingCell(cell, dt, t, T_floor, gamma);
		U[4 * n + i] = cell[4];
		U[5 * n + i] = cell[5];
	}
}

} // extern "C"
