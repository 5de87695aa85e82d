// quokka_advection.hpp — the scalar linear-advection solver of the reference on this host mirror:
//   LinearAdvectionSystem<problem_t>   reference src/linear_advection/linear_advection.hpp        -> qk_advect_* of the C-ABI (+ qk_ReconstructStatesPPM)
//   AdvectionSimulation<problem_t>     reference src/linear_advection/AdvectionSimulation.hpp     -> the level-0 driver below
// so that src/problems/Advection, AdvectionSemiellipse and Advection2D compile unchanged.  With amr.max_level > 0 the levels are objects of this
// class under the level machinery of quokka_amr.hpp (AmrDriver<problem_t, AdvectionSimulation<problem_t>>: FillPatch without energy hooks, both
// RK stages added to the flux registers with half the step): Advection2D's ctest deck refines three levels and meets its 0.15 criterion that
// way (0.1447; 0.34 on level 0 alone — docs/DESIGN_ROUNDS_1_4.md §17).
#ifndef QK_HOST_QUOKKA_ADVECTION_HPP_
#define QK_HOST_QUOKKA_ADVECTION_HPP_

#include "quokka_amr.hpp"

template <typename problem_t> class LinearAdvectionSystem : public HyperbolicSystem<problem_t>
{
      public:
	enum varIndex { density_index = 0 };
	static auto lev() -> qk_level * { return qkhost::Runtime::get().lev; }
	// linear_advection.hpp:62-70: the primitive variable IS the conserved one
	static void ConservedToPrimitive(amrex::MultiFab const &cons_mf, amrex::MultiFab &primVar_mf, int /*nghost*/, int nvars)
	{
		amrex::MultiFab::Copy(primVar_mf, cons_mf, 0, 0, nvars, primVar_mf.nGrow());
	}
	// :165-198
	template <FluxDir DIR>
	static void ComputeFluxes(amrex::MultiFab &x1Flux_mf, amrex::MultiFab const &x1LeftState_mf, amrex::MultiFab const &x1RightState_mf, double advectionVx, int nvars)
	{
		qkhost::check(qk_advect_ComputeFluxes(lev(), nullptr, static_cast<int>(DIR), qkhost::tab(x1Flux_mf), qkhost::tab(x1LeftState_mf), qkhost::tab(x1RightState_mf),
						      advectionVx, nvars),
			      "LinearAdvectionSystem::ComputeFluxes");
	}
	static void flux3(std::array<amrex::MultiFab, AMREX_SPACEDIM> const &f, const qk_array4 *out[3])
	{
		for (int d = 0; d < 3; ++d) {
			out[d] = (d < AMREX_SPACEDIM) ? qkhost::tab(f[d]) : nullptr;
		}
	}
	static void dx3(amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &dx, double out[3])
	{
		for (int d = 0; d < 3; ++d) {
			out[d] = (d < AMREX_SPACEDIM) ? dx[d] : 1.0;
		}
	}
	// :82-118
	static void PredictStep(amrex::MultiFab const &consVarOld_mf, amrex::MultiFab &consVarNew_mf, std::array<amrex::MultiFab, AMREX_SPACEDIM> const &fluxArray, double dt,
				amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> dx_in, int nvars)
	{
		const qk_array4 *f[3];
		double d3[3];
		flux3(fluxArray, f);
		dx3(dx_in, d3);
		qkhost::check(qk_advect_PredictStep(lev(), nullptr, qkhost::tab(consVarOld_mf), qkhost::tab(consVarNew_mf), f, dt, d3, nvars), "LinearAdvectionSystem::PredictStep");
	}
	// :120-163
	static void AddFluxesRK2(amrex::MultiFab &U_new_mf, amrex::MultiFab const &U0_mf, amrex::MultiFab const &U1_mf,
				 std::array<amrex::MultiFab, AMREX_SPACEDIM> const &fluxArray, double dt, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> dx_in, int nvars)
	{
		const qk_array4 *f[3];
		double d3[3];
		flux3(fluxArray, f);
		dx3(dx_in, d3);
		qkhost::check(qk_advect_AddFluxesRK2(lev(), nullptr, qkhost::tab(U_new_mf), qkhost::tab(U0_mf), qkhost::tab(U1_mf), f, dt, d3, nvars),
			      "LinearAdvectionSystem::AddFluxesRK2");
	}
};

template <typename problem_t> class AdvectionSimulation : public AMRSimulation<problem_t>
{
      public:
	using AMRSimulation<problem_t>::state_old_cc_;
	using AMRSimulation<problem_t>::state_new_cc_;
	using AMRSimulation<problem_t>::cflNumber_;
	using AMRSimulation<problem_t>::dt_;
	using AMRSimulation<problem_t>::tNew_;
	using AMRSimulation<problem_t>::istep;
	using AMRSimulation<problem_t>::geom;
	using AMRSimulation<problem_t>::grids_;
	using AMRSimulation<problem_t>::nghost_cc_;
	using AMRSimulation<problem_t>::stopTime_;
	using AMRSimulation<problem_t>::maxTimesteps_;
	using AMRSimulation<problem_t>::componentNames_cc_;
	using AMRSimulation<problem_t>::fillBoundaryConditions;
	using AMRSimulation<problem_t>::boxArray;
	using AMRSimulation<problem_t>::DistributionMap;

	explicit AdvectionSimulation(amrex::Vector<amrex::BCRec> &BCs_cc) : AMRSimulation<problem_t>(BCs_cc) { allocate(); }
	// one level of a hierarchy (quokka_amr.hpp)
	AdvectionSimulation(amrex::Vector<amrex::BCRec> &BCs_cc, LevelSpec const &spec) : AMRSimulation<problem_t>(BCs_cc, spec) { allocate(); }

	// --- the level interface of AmrDriver (quokka_amr.hpp)
	static constexpr bool isAdvection = true;
	static constexpr int ncompHydro_ = Physics_Indices<problem_t>::nvarTotal_cc; // components of the flux registers
	double tOldLev_ = 0.0, tNewLev_ = 0.0;
	double fillTime_ = 0.0;
	[[nodiscard]] auto bcFillTime() const -> double override { return fillTime_; }
	double elapsedSeconds_ = 0.0;
	std::function<void(std::array<amrex::MultiFab, AMREX_SPACEDIM> &, double)> afterStageFluxes_; // incrementFluxRegisters(fluxArrays, 0.5 dt) of a stage
	void inheritSettings(AdvectionSimulation const &base)
	{
		cflNumber_ = base.cflNumber_;
		advectionVx_ = base.advectionVx_;
		advectionVy_ = base.advectionVy_;
		advectionVz_ = base.advectionVz_;
		this->constantDt_ = base.constantDt_;
	}
	void FixupState() {}
	void dropCachedSignal() {}
	// CFL time step of this level alone: LinearAdvectionSystem::ComputeMaxSignalSpeed (linear_advection.hpp:47-60) is the same number in every cell
	[[nodiscard]] auto computeTimestepAtLevel() const -> double
	{
		double const signal = std::sqrt(advectionVx_ * advectionVx_ + advectionVy_ * advectionVy_ + advectionVz_ * advectionVz_);
		double dx_min = geom[0].dx[0];
		for (int d = 1; d < AMREX_SPACEDIM; ++d) {
			dx_min = std::min(dx_min, geom[0].dx[d]);
		}
		return cflNumber_ * (dx_min / signal);
	}
	// advanceSingleTimestepAtLevel for a level of a hierarchy: the ghost cells of a refined level come from its parent at `time` (stage 1) and
	// `time + dt` (stage 2) — AdvectionSimulation.hpp:294, :326
	auto advanceLevel(double time, double dt_lev) -> bool
	{
		levelTime_ = time;
		advanceSingleTimestepAtLevel(0, time, dt_lev, 1);
		return true;
	}

      private:
	double levelTime_ = 0.0;
	void allocate()
	{
		componentNames_cc_.push_back({"density"});
		int const nc = Physics_Indices<problem_t>::nvarTotal_cc;
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			flux_[d].define(grids_, nc, 0, d);
			leftState_[d].define(grids_, nc, 1, d);
			rightState_[d].define(grids_, nc, 1, d);
		}
		primVar_.define(grids_, nc, nghost_cc_);
	}

      public:
	// the reference's data members (AdvectionSimulation.hpp:89-96)
	double advectionVx_ = 1.0;
	double advectionVy_ = 0.0;
	double advectionVz_ = 0.0;
	amrex::Real errorNorm_ = std::numeric_limits<double>::quiet_NaN();
	static constexpr int reconstructOrder_ = 3;
	static constexpr int integratorOrder_ = 2;

	void setInitialConditionsOnGrid(quokka::grid const &grid_elem) override;
	void computeReferenceSolution(amrex::MultiFab &ref, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &dx, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &prob_lo,
				      amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const &prob_hi);
	void ErrorEst(int lev, amrex::TagBoxArray &tags, amrex::Real time, int ngrow);

	// AdvectionSimulation.hpp:385-434
	template <FluxDir DIR> void fluxFunction(amrex::MultiFab const &consState, int nvars)
	{
		constexpr int d = static_cast<int>(DIR);
		double const vel = (d == 0) ? advectionVx_ : (d == 1) ? advectionVy_ : advectionVz_;
		LinearAdvectionSystem<problem_t>::ConservedToPrimitive(consState, primVar_, nghost_cc_, nvars);
		LinearAdvectionSystem<problem_t>::template ReconstructStatesPPM<DIR>(primVar_, leftState_[d], rightState_[d], 1, nvars);
		LinearAdvectionSystem<problem_t>::template ComputeFluxes<DIR>(flux_[d], leftState_[d], rightState_[d], vel, nvars);
	}
	auto computeFluxes(amrex::MultiFab const &consVar, int nvars, int /*lev*/) -> std::array<amrex::MultiFab, AMREX_SPACEDIM> &
	{
		AMREX_D_TERM(fluxFunction<FluxDir::X1>(consVar, nvars);, fluxFunction<FluxDir::X2>(consVar, nvars);, fluxFunction<FluxDir::X3>(consVar, nvars);)
		return flux_;
	}

	// AdvectionSimulation.hpp:236-383 on one level (integratorOrder_ = 2)
	void advanceSingleTimestepAtLevel(int lev, amrex::Real /*time*/, amrex::Real dt_lev, int /*ncycle*/)
	{
		int const nvars = Physics_Indices<problem_t>::nvarTotal_cc;
		double const fluxScaleFactor = 0.5; // integratorOrder_ == 2 (:297-302)
		this->activate();
		std::swap(state_old_cc_[lev], state_new_cc_[lev]);
		fillTime_ = levelTime_;
		fillBoundaryConditions(state_old_cc_[lev]);
		{
			auto &fluxArrays = computeFluxes(state_old_cc_[lev], nvars, lev);
			LinearAdvectionSystem<problem_t>::PredictStep(state_old_cc_[lev], state_new_cc_[lev], fluxArrays, dt_lev, geom[lev].CellSizeArray(), nvars);
			if (afterStageFluxes_) {
				afterStageFluxes_(fluxArrays, fluxScaleFactor * dt_lev);
			}
		}
		fillTime_ = levelTime_ + dt_lev;
		fillBoundaryConditions(state_new_cc_[lev]);
		{
			auto &fluxArrays = computeFluxes(state_new_cc_[lev], nvars, lev);
			LinearAdvectionSystem<problem_t>::AddFluxesRK2(state_new_cc_[lev], state_old_cc_[lev], state_new_cc_[lev], fluxArrays, dt_lev, geom[lev].CellSizeArray(), nvars);
			if (afterStageFluxes_) {
				afterStageFluxes_(fluxArrays, fluxScaleFactor * dt_lev);
			}
		}
	}

	// dt: LinearAdvectionSystem::ComputeMaxSignalSpeed (linear_advection.hpp:47-60) is the same number in every cell; AMRSimulation::computeTimestep
	// (reference src/simulation.hpp:722-818, single level)
	void computeTimestep()
	{
		double dt_tmp = computeTimestepAtLevel();
		dt_tmp = std::min(dt_tmp, 1.1 * dt_[0]);
		double dt_0 = std::min(dt_tmp, 1.0 * dt_tmp);
		dt_0 = std::min(dt_0, this->maxDt_);
		if (tNew_[0] == 0.0) {
			dt_0 = std::min(dt_0, this->initDt_);
		}
		if (this->constantDt_ > 0.0) {
			dt_0 = this->constantDt_;
		}
		double const eps = 1.e-3 * dt_0;
		if (tNew_[0] + dt_0 > stopTime_ - eps) {
			dt_0 = stopTime_ - tNew_[0];
		}
		dt_[0] = dt_0;
	}

	// AdvectionSimulation.hpp:196-222
	void computeAfterEvolve(amrex::Vector<amrex::Real> & /*initSumCons*/) override
	{
		int const ncomp = state_new_cc_[0].nComp();
		amrex::MultiFab ref(grids_, ncomp, 0);
		computeReferenceSolution(ref, geom[0].CellSizeArray(), geom[0].ProbLoArray(), geom[0].ProbHiArray());
		double sol_norm = 0., err_norm = 0.;
		for (int n = 0; n < ncomp; ++n) {
			double rn = 0., en = 0.;
			for (int b = 0; b < ref.size(); ++b) {
				auto hr = ref.copyToHost(b);
				auto hs = state_new_cc_[0].copyToHost(b);
				amrex::Array4<double> r(hr.data(), ref.fabbox(b), ncomp);
				amrex::Array4<double> s(hs.data(), state_new_cc_[0].fabbox(b), ncomp);
				amrex::HostFor(ref.validbox(b), [&](int i, int j, int k) {
					rn += std::abs(r(i, j, k, n));
					en += std::abs(r(i, j, k, n) - s(i, j, k, n));
				});
			}
			rn = qkhost::Comm::get().allReduceSum(rn);
			en = qkhost::Comm::get().allReduceSum(en);
			sol_norm += rn * rn;
			err_norm += en * en;
		}
		errorNorm_ = std::sqrt(err_norm) / std::sqrt(sol_norm);
		amrex::Print() << "\nRelative rms L1 error norm = " << errorNorm_ << "\n\n";
	}

	// AMRSimulation::setInitialConditions: a hierarchy when the deck asks for one
	void setInitialConditions()
	{
		int max_level = 0;
		amrex::ParmParse("amr").query("max_level", max_level);
		if (max_level > 0) {
			amr_ = std::make_shared<AmrDriver<problem_t, AdvectionSimulation<problem_t>>>(*this);
			amr_->setInitialConditions();
		} else {
			AMRSimulation<problem_t>::setInitialConditions();
		}
	}
	std::shared_ptr<AmrDriver<problem_t, AdvectionSimulation<problem_t>>> amr_;

	// reference src/simulation.hpp:856-951
	void evolve()
	{
		AMREX_ALWAYS_ASSERT(this->areInitialConditionsDefined_);
		if (amr_) {
			amr_->evolve(); // (ends with computeAfterEvolve: the level-0 state holds the average of every finer level)
			qkDumpFields(state_new_cc_[0], istep[0], tNew_[0], dt_[0], 0, 0, errorNorm_);
			return;
		}
		amrex::Vector<amrex::Real> init_sum_cons(1, 0.0);
		QK_HOST_HIP(hipDeviceSynchronize());
		auto const t0 = std::chrono::steady_clock::now();
		double cur_time = tNew_[0];
		for (int step = istep[0]; step < maxTimesteps_ && cur_time < stopTime_; ++step) {
			computeTimestep();
			double const time = tNew_[0];
			advanceSingleTimestepAtLevel(0, time, dt_[0], 1);
			++istep[0];
			this->cellUpdates_ += this->CountCells(0);
			cur_time += dt_[0];
			tNew_[0] = cur_time;
			if (cur_time >= stopTime_ - 1.e-6 * dt_[0]) {
				break;
			}
		}
		QK_HOST_HIP(hipDeviceSynchronize());
		double const elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		amrex::Print() << "elapsed time: " << elapsed << " seconds.\n";
		amrex::Print() << "Performance figure-of-merit: " << 1.0e6 * elapsed / static_cast<double>(this->cellUpdates_) << " microseconds per zone-update ["
			       << static_cast<double>(this->cellUpdates_) / elapsed / 1.0e6 << " Mupdates/s]\n";
		computeAfterEvolve(init_sum_cons);
		qkDumpFields(state_new_cc_[0], istep[0], tNew_[0], dt_[0], 0, 0, errorNorm_); // test hook `qk.dump_state=<file>`
	}

      private:
	std::array<amrex::MultiFab, AMREX_SPACEDIM> flux_, leftState_, rightState_;
	amrex::MultiFab primVar_;
};

// the hooks a problem may leave alone (AdvectionSimulation.hpp:143-194)
template <typename problem_t> void AdvectionSimulation<problem_t>::setInitialConditionsOnGrid(quokka::grid const & /*grid_elem*/) {}
template <typename problem_t>
void AdvectionSimulation<problem_t>::computeReferenceSolution(amrex::MultiFab & /*ref*/, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const & /*dx*/,
							      amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const & /*prob_lo*/,
							      amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const & /*prob_hi*/)
{
}
template <typename problem_t> void AdvectionSimulation<problem_t>::ErrorEst(int /*lev*/, amrex::TagBoxArray & /*tags*/, amrex::Real /*time*/, int /*ngrow*/) {}

#endif // QK_HOST_QUOKKA_ADVECTION_HPP_
