// quokka_host.hpp — C++17 host mirror of the reference's operator surface on the hydro and radiation path, in four parts:
//   quokka_hydro_system.hpp     constants, Physics_Traits / Physics_Indices, quokka::EOS, HyperbolicSystem<problem_t>, HydroSystem<problem_t>
//   quokka_rad_system.hpp       RadSystem_Traits, RadSystem<problem_t>
//   quokka_amr_simulation.hpp   AMRSimulation<problem_t>: decks, geometry, arrays, ghost fill, output, problem hooks
//   this file                   QuokkaSimulation<problem_t>: the evolve loop and dt control (reference src/simulation.hpp:703-981), and THE HOT-PATH SCHEDULE —
//                               advanceHydroAtLevelWithRetries / advanceHydroAtLevel (RK2 + first-order flux correction + retries, reference
//                               src/QuokkaSimulation.hpp:885-1322), the fused stage pair with one read-back per step (stagePairSpeculative), the early / late
//                               ghost-exchange overlap (launchStage), the radiation subcycle (:1570-1961).
// Problems specialise the same trait structs and member templates as in the reference (setInitialConditionsOnGrid, setCustomBoundaryConditions,
// computeAfterEvolve, ...); their device lambdas and hook functions are compiled by hipcc in the problem's translation unit and run as kernels.
#ifndef QK_HOST_QUOKKA_HOST_HPP_
#define QK_HOST_QUOKKA_HOST_HPP_

#include "quokka_amr_simulation.hpp"
#include "compat/grackle_like_cooling.hpp"
#include "compat/tabulated_cooling.hpp"

template <typename problem_t> class QuokkaSimulation : public AMRSimulation<problem_t>
{
      public:
	using B = AMRSimulation<problem_t>;
	using B::BCs_cc_;
	using B::cflNumber_;
	using B::densityFloor_;
	using B::dt_;
	using B::geom;
	using B::grids_;
	using B::istep;
	using B::maxTimesteps_;
	using B::nghost_cc_;
	using B::state_new_cc_;
	using B::state_old_cc_;
	using B::stopTime_;
	using B::tempFloor_;
	using B::tNew_;

	// reference src/QuokkaSimulation.hpp:125-144
	int integratorOrder_ = 2;
	int reconstructionOrder_ = 3;
	int radiationReconstructionOrder_ = 3;
	int useDualEnergy_ = 1;
	int abortOnFofcFailure_ = 1;
	amrex::Real artificialViscosityK_ = 0.;
	bool computeReferenceSolution_ = false;
	amrex::Real errorNorm_ = std::numeric_limits<double>::quiet_NaN();
	amrex::Real pressureFloor_ = 0.;
	// cooling (reference src/QuokkaSimulation.hpp:110-118)
	int enableCooling_ = 0;
	quokka::GrackleLikeCooling::grackle_tables grackleTables_; // (declared only: compat/grackle_like_cooling.hpp)
	quokka::TabulatedCooling::cloudy_tables cloudyTables_;
	std::string coolingTableType_{};
	std::string coolingTableFilename_{};
	long fofcStages_ = 0, retries_ = 0;
	double elapsedSeconds_ = 0.0;
	// radiation (reference src/QuokkaSimulation.hpp:127-131)
	amrex::Real radiationCflNumber_ = 0.3;
	amrex::Real dustGasInteractionCoeff_ = 2.5e-34; // erg cm^3 s^-1 K^-3/2 (QuokkaSimulation.hpp:127; radiation.dust_gas_interaction_coeff, :392)
	int maxSubsteps_ = 10;
	bool afterTimestepIsDefault_ = false, beforeTimestepIsDefault_ = false; // set by the default computeAfterTimestep / computeBeforeTimestep
	bool radSourceTimeIndependent_ = false; // deck: radiation.source_is_time_independent (fillRadEnergySource)
	amrex::Long radiationCellUpdates_ = 0;
	long radSolves_ = 0, radNewtonIterations_ = 0;
	int radMaxNewtonIterations_ = 0;
	static constexpr bool is_radiation_enabled_ = Physics_Traits<problem_t>::is_radiation_enabled;

	static constexpr int ncompHydro_ = HydroSystem<problem_t>::nvar_;

	explicit QuokkaSimulation(amrex::Vector<amrex::BCRec> &BCs_cc) : AMRSimulation<problem_t>(BCs_cc) { construct(); }
	QuokkaSimulation(amrex::Vector<amrex::BCRec> &BCs_cc, amrex::Vector<amrex::BCRec> &BCs_fc) : AMRSimulation<problem_t>(BCs_cc, BCs_fc) { construct(); }
	// one level of an AMR hierarchy (quokka_amr.hpp)
	QuokkaSimulation(amrex::Vector<amrex::BCRec> &BCs_cc, LevelSpec const &spec) : AMRSimulation<problem_t>(BCs_cc, spec) { construct(); }

	// --- AMR level bookkeeping (used by quokka_amr.hpp; a uniform-grid run leaves the defaults)
	static constexpr bool isAdvection = false;
	// what the problem set on the level-0 object in problem_main, handed to the object of a refined level
	void inheritSettings(QuokkaSimulation const &base)
	{
		cflNumber_ = base.cflNumber_;
		densityFloor_ = base.densityFloor_;
		tempFloor_ = base.tempFloor_;
		reconstructionOrder_ = base.reconstructionOrder_;
		integratorOrder_ = base.integratorOrder_;
		useDualEnergy_ = base.useDualEnergy_;
		abortOnFofcFailure_ = base.abortOnFofcFailure_;
		artificialViscosityK_ = base.artificialViscosityK_;
		radiationCflNumber_ = base.radiationCflNumber_;
		radiationReconstructionOrder_ = base.radiationReconstructionOrder_;
		use_wavespeed_correction_ = base.use_wavespeed_correction_;
		maxSubsteps_ = base.maxSubsteps_;
		radSourceTimeIndependent_ = base.radSourceTimeIndependent_;
		dustGasInteractionCoeff_ = base.dustGasInteractionCoeff_;
		this->constantDt_ = base.constantDt_;
	}
	double tOldLev_ = 0.0, tNewLev_ = 0.0; // tOld_[lev], tNew_[lev]
	bool use_wavespeed_correction_ = false; // QuokkaSimulation.hpp:133 (ComputeCellOpticalDepth + S_corr on the even faces: wavespeedEps below)
	double fillTime_ = 0.0;		       // the time a ghost fill refers to (coarse data are interpolated to it)
	[[nodiscard]] auto bcFillTime() const -> double override { return fillTime_; }
	bool storeFluxRk2_ = false;	       // keep flux_rk2 = 0.5 F1 + 0.5 F2 (rk2flux_) for the flux registers
	// The carried form on a level with refined children (hydro.rk2_carry_rhs = 1 on the base level of a hierarchy): flux_rk2 is formed only on the faces
	// of the cells the child's flux register marks (qk_hydro_stage_args::flux_mask); set by AmrDriver when it links level 1 (quokka_amr.hpp)
	amrex::TagBoxArray fluxMask_;
	[[nodiscard]] auto needsFluxRk2() const -> bool { return storeFluxRk2_ || fluxMask_.size() > 0; }
	[[nodiscard]] auto wantsCarriedForm() const -> bool { return AMREX_SPACEDIM == 3 && rk2CarryRhs_ != 0 && integratorOrder_ == 2; }
	void setFluxMaskFrom(qk_fluxreg *reg)
	{
		fluxMask_.define(grids_, 1, 1);
		std::vector<std::vector<char>> h(static_cast<size_t>(fluxMask_.size()));
		for (int b = 0; b < fluxMask_.size(); ++b) {
			h[b].assign(static_cast<size_t>(fluxMask_.fabbox(b).numPts()), 0);
		}
		std::vector<amrex::Box> win(static_cast<size_t>(fluxMask_.size())); // bounding box of the marked cells of every box
		for (auto &w : win) {
			w.hi[0] = w.hi[1] = w.hi[2] = -1; // (empty until an item marks a cell: a default Box is the cell (0, 0, 0))
		}
		for (int n = 0; n < qk_fluxreg_num_items(reg); ++n) {
			int dir = 0, side = 0, fb = 0, cb = 0, lo[3], hi[3], sh[3];
			qkhost::check(qk_fluxreg_item(reg, n, &dir, &side, &fb, &cb, lo, hi, sh), "qk_fluxreg_item");
			amrex::Array4<char> a(h[cb].data(), fluxMask_.fabbox(cb), 1);
			amrex::Box &w = win[static_cast<size_t>(cb)];
			bool const first = !w.ok();
			for (int d = 0; d < 3; ++d) {
				w.lo[d] = first ? lo[d] + sh[d] : std::min(w.lo[d], lo[d] + sh[d]);
				w.hi[d] = first ? hi[d] + sh[d] : std::max(w.hi[d], hi[d] + sh[d]);
			}
			for (int k = lo[2]; k <= hi[2]; ++k) {
				for (int j = lo[1]; j <= hi[1]; ++j) {
					for (int i = lo[0]; i <= hi[0]; ++i) {
						a(i + sh[0], j + sh[1], k + sh[2]) = 1;
					}
				}
			}
		}
		for (int b = 0; b < fluxMask_.size(); ++b) {
			fluxMask_.copyFromHost(b, h[b]);
		}
		fluxMask_.cropDeviceTable(win); // (the kernels drop a face outside the window without reading a byte: include/quokka_amd.h, flux_mask)
		for (auto &g : groups_) { // descriptors gathered per box group from an earlier mask at the same address are stale
			auto it = g.tables.find(static_cast<const void *>(fluxMask_.arrays()));
			if (it != g.tables.end()) {
				(void)hipFree(it->second);
				g.tables.erase(it);
			}
		}
		storeFluxRk2_ = false;
	}
	std::function<void(double)> afterAdvance_; // incrementFluxRegisters(dt) after every successful advanceHydroAtLevel
	// a level of a hierarchy with radiation: the radiation fluxes of a stage go to the flux registers of the radiation block
	std::function<void(std::array<amrex::MultiFab, AMREX_SPACEDIM> &, double)> afterRadStage_;
	double radTime_ = 0.0; // start of the current radiation substep
	std::function<void(int)> beforeAttempt_;   // flux registers: save before the retry loop (0), back to that state at every retry (>0)
	// what the flux registers accumulate after a level advance (reference src/QuokkaSimulation.hpp:1303-1306)
	[[nodiscard]] auto halfFlux() -> std::array<amrex::MultiFab, AMREX_SPACEDIM> & { return (integratorOrder_ == 2) ? rk2flux_ : halfFlux_; }
	// ErrorEst(lev, tags, time, ngrow): problem hook (reference src/QuokkaSimulation.hpp:213).  Device lambdas cannot be compiled
	// against the C-ABI; the gradient-threshold family of the reference's problems is one library call: tagRelativeGradient below.
	virtual void ErrorEst(int /*lev*/, amrex::TagBoxArray & /*tags*/, amrex::Real /*time*/, int /*ngrow*/) {}
	// field: QK_TAGFIELD_PRESSURE or a component index;  tags SET where max_d max(|q+ - q|, |q - q-|) / q > eta and q > qmin (>= if inclusive)
	void tagRelativeGradient(amrex::TagBoxArray &tags, int field, double eta_threshold, double q_min, bool min_inclusive)
	{
		this->activate();
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_tag_relative_gradient(this->levelHandle(), nullptr, &t, qkhost::tab(state_new_cc_[0]), reinterpret_cast<qk_carray4 *>(tags.arrays()), field,
						       eta_threshold, q_min, min_inclusive ? 1 : 0),
			      "qk_tag_relative_gradient");
	}
	// centred-difference form (HydroShocktube): SET where sqrt(((q(+1) - q(-1)) / (2 dx))^2) / q > eta and q >= qmin (> if !inclusive)
	// (dx <= 0: the level's cell size, as HydroShocktube divides by it; PassiveScalar's form has no dx: pass 1)
	void tagCenteredGradient(amrex::TagBoxArray &tags, int comp, int dir, double eta_threshold, double q_min, bool min_inclusive, double dx = -1.0)
	{
		this->activate();
		qkhost::check(qk_tag_centered_gradient(this->levelHandle(), nullptr, qkhost::tab(state_new_cc_[0]), reinterpret_cast<qk_carray4 *>(tags.arrays()), comp, dir,
						       dx > 0.0 ? dx : geom[0].dx[dir], eta_threshold, q_min, min_inclusive ? 1 : 0),
			      "qk_tag_centered_gradient");
	}
	void FixupState() // reference src/QuokkaSimulation.hpp:761-770
	{
		if constexpr (!Physics_Traits<problem_t>::is_hydro_enabled) {
			return;
		}
		this->activate();
		// EnforceLimits + SyncDualEnergy in one pass that also reduces the CFL maxima of the result (qk_hydro_FixupState); they stay on the device
		// until the next time step asks for them (resolveSignal)
		if (d_fixSignal_ == nullptr) {
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_fixSignal_), 2 * sizeof(double)));
		}
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_hydro_FixupState(this->levelHandle(), nullptr, &t, densityFloor_, tempFloor_, useDualEnergy_, qkhost::tab(state_new_cc_[0]), d_error_,
						  d_fixSignal_),
			      "qk_hydro_FixupState");
		invalidateSignal();
		signalPending_ = true;
	}
	void invalidateSignal()
	{
		haveSignal_ = false;
		signalPending_ = false;
		fixFarPending_ = false;
	}
	// the CFL maxima FixupState left on the device, over all ranks (every rank calls this at the same points: computeTimestepAtLevel)
	void resolveSignal()
	{
		if (!haveSignal_ && signalPending_) {
			QK_HOST_HIP(hipMemcpy(signal_, d_fixSignal_, 2 * sizeof(double), hipMemcpyDeviceToHost));
			if (fixFarPending_) { // the level was fixed up in two parts (fixupNear): the maxima of the other part
				double far[2];
				QK_HOST_HIP(hipMemcpy(far, d_fixFar_, 2 * sizeof(double), hipMemcpyDeviceToHost));
				signal_[0] = std::max(signal_[0], far[0]);
				signal_[1] = std::max(signal_[1], far[1]);
				fixFarPending_ = false;
			}
			if (qkhost::Comm::get().size > 1) {
				qkhost::Comm::get().allReduce(signal_, 2, qkhost::Comm::Op::max);
			}
			haveSignal_ = true;
			signalPending_ = false;
		}
	}
	// CFL time step of this level alone (reference src/simulation.hpp:703-720)
	[[nodiscard]] auto computeTimestepAtLevel() -> double
	{
		this->activate();
		double m = 0.0;
		if constexpr (!Physics_Traits<problem_t>::is_hydro_enabled) { // radiation only (reference src/QuokkaSimulation.hpp:421-424)
			m = RadSystem<problem_t>::c_hat_;
		} else {
			resolveSignal();
			m = (haveSignal_ ? signal_[1] : HydroSystem<problem_t>::maxSignalSpeedLocal(state_new_cc_[0], 1));
		}
		if constexpr (is_radiation_enabled_ && Physics_Traits<problem_t>::is_hydro_enabled) { // :421-434
			m = std::max(RadSystem<problem_t>::c_hat_ / static_cast<double>(maxSubsteps_), m);
		}
		return cflNumber_ * (minDx() / m);
	}
	// ------------------------------------------------------------------------------------------------------------------------------------------
	// A coarse step whose verdict is read AFTER its children have been advanced (AmrDriver::timeStepWithSubcycling; quokka_amd/amr_simulation.py
	// has the same schedule).  Level 0 of a young hierarchy holds 8 boxes of which one is refined: stage 2 of the 7 boxes no child reads ("far")
	// runs on a second stream with a scratch array of its own while the children — chains of 20-60 us kernels that leave the chip nearly empty
	// — are advanced on the compute stream.  Level 0's redo counts and CFL maxima then arrive after the children: they run speculatively, and
	// so do their own level steps (advanceLevelDeferred: the eight words a step reports in go to a device log, read once per coarse step).
	// A bad verdict anywhere rolls the coarse step back (AmrDriver) and redoes it in the ordinary order.
	[[nodiscard]] auto canSpeculate() const -> bool
	{
		if constexpr (!fusedEligible() || is_radiation_enabled_ || !Physics_Traits<problem_t>::is_hydro_enabled || AMREX_SPACEDIM != 3) {
			return false;
		} else {
			return integratorOrder_ == 2 && speculateStage2_ != 0 && strangSourcesAreDefault_ && afterLevelAdvanceIsDefault_ && enableCooling_ == 0 &&
			       this->customBcIsDefault_ == 1;
		}
	}
	// a level step inside a speculative coarse step: both stages and the register increments enqueued, the words copied to `d_log` (8 x int64)
	void advanceLevelDeferred(double time, double dt_lev, int64_t *d_log)
	{
		std::swap(state_old_cc_[0], state_new_cc_[0]);
		this->activate();
		invalidateSignal();
		if (beforeAttempt_) {
			beforeAttempt_(0);
		}
		fusedBegin(1, true);
		fillTime_ = time;
		launchStage(1, state_old_cc_[0], state_old_cc_[0], state_inter_cc_, dt_lev, 0);
		fillTime_ = time + dt_lev;
		launchStage(2, state_inter_cc_, state_old_cc_[0], state_new_cc_[0], dt_lev, 1);
		QK_HOST_HIP(hipMemcpyAsync(d_log, d_words_, 8 * sizeof(int64_t), hipMemcpyDeviceToDevice, qkhost::Runtime::get().computeStream()));
		stage1LeftF1_ = !carryActive();
		if (afterAdvance_) {
			afterAdvance_(dt_lev);
		}
		this->oldStateGhostsFilled_ = true; // (stage 1 filled the old state's ghost cells in place, at `time`)
	}
	// what a deferred step's log entry said (AmrDriver::deferredVerdicts): the CFL maxima of the final stage, kept if nothing has changed the state since
	void adoptDeferredSignal(double sig0, double sig1)
	{
		if (!signalPending_ && !haveSignal_) {
			signal_[0] = sig0;
			signal_[1] = sig1;
			haveSignal_ = true;
		}
	}
	[[nodiscard]] auto cflLimitFor(double max_signal) const -> double { return cflNumber_ * (minDx() / max_signal); }

	// near: the boxes the child level reads (AmrDriver::speculativeSplit); far: the others, as up to eight launch sets (each holds its share of the
	// wave slots only: the children's small kernels find free slots sooner — profiles/round5/ab9_amr_far_split.txt)
	void setSpeculativeSplit(std::vector<int> const &near, std::vector<int> const &far)
	{
		auto make = [&](OverlapGroup &G, std::vector<int> const &idx) {
			releaseGroup(G);
			G.idx = idx;
			std::vector<qk_box> qb;
			for (int b : idx) {
				qb.push_back({{grids_[b].lo[0], grids_[b].lo[1], grids_[b].lo[2]}, {grids_[b].hi[0], grids_[b].hi[1], grids_[b].hi[2]}});
			}
			qkhost::check(qk_level_create(qkhost::Runtime::get().ctx, &G.lev, AMREX_SPACEDIM, static_cast<int>(qb.size()), qb.data()), "qk_level_create");
		};
		make(specNear_, near);
		make(specFar_, far);
		for (auto &G : specFarParts_) {
			releaseGroup(G);
		}
		int nsplit = 8;
		amrex::ParmParse("qk").query("amr_far_split", nsplit);
		nsplit = std::max(1, std::min<int>(static_cast<int>(far.size()), nsplit));
		specFarParts_.assign(static_cast<size_t>(nsplit), OverlapGroup{});
		for (int k = 0; k < nsplit; ++k) {
			std::vector<int> part;
			for (size_t n = static_cast<size_t>(k); n < far.size(); n += static_cast<size_t>(nsplit)) {
				part.push_back(far[n]);
			}
			make(specFarParts_[static_cast<size_t>(k)], part);
		}
		auto t = qkhost::traits<problem_t>();
		int64_t const need = qk_hydro_stage_scratch_bytes(specFar_.lev, &t);
		if (need > farScratchBytes_) {
			(void)hipFree(farScratch_);
			QK_HOST_HIP(hipMalloc(&farScratch_, static_cast<size_t>(need)));
			farScratchBytes_ = need;
		}
		if (farStream_ == nullptr) {
			int least = 0, greatest = 0;
			QK_HOST_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
			QK_HOST_HIP(hipStreamCreateWithPriority(&farStream_, hipStreamNonBlocking, least)); // the far boxes fill what the children leave idle
			QK_HOST_HIP(hipEventCreateWithFlags(&evInterReady_, hipEventDisableTiming));
			QK_HOST_HIP(hipEventCreateWithFlags(&evFarDone_, hipEventDisableTiming));
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_fixFar_), 2 * sizeof(double)));
		}
		std::vector<char> second(grids_.size(), 0);
		for (int b : far) {
			second[static_cast<size_t>(b)] = 1;
		}
		this->setBoxGroups(second);
	}
	// advanceLevel with the verdict deferred: ghost fill + stage 1 of all boxes, ghost fill + stage 2 of the near boxes on the compute stream, stage 2
	// (and FixupState) of the far boxes on the side stream; the physical boundaries of the near boxes' new state and the coarse side of the child's
	// flux register follow on the compute stream, so that the children can be enqueued at once.  Nothing is read back.
	void advanceLevelBegin(double time, double dt_lev)
	{
		std::swap(state_old_cc_[0], state_new_cc_[0]);
		this->oldStateGhostsFilled_ = false;
		this->activate();
		invalidateSignal();
		hipStream_t const cs = qkhost::Runtime::get().computeStream();
		if (beforeAttempt_) {
			beforeAttempt_(0);
		}
		primNow_ = primBackoff_ == 0 && primHandoffApplies();
		fusedBegin(1, true);
		fillTime_ = time;
		this->fillBoundaryConditions(state_old_cc_[0]);
		fusedLaunch(1, state_old_cc_[0], state_old_cc_[0], state_inter_cc_, dt_lev, -1, false, 0);
		fillTime_ = time + dt_lev;
		this->fillBoundaryConditions(state_inter_cc_);
		QK_HOST_HIP(hipEventRecord(evInterReady_, cs));
		fusedLaunch(2, state_inter_cc_, state_old_cc_[0], state_new_cc_[0], dt_lev, -1, false, 1, &specNear_);
		QK_HOST_HIP(hipStreamWaitEvent(farStream_, evInterReady_, 0));
		for (auto &part : specFarParts_) {
			fusedLaunch(2, state_inter_cc_, state_old_cc_[0], state_new_cc_[0], dt_lev, -1, false, 1, &part, farScratch_, farScratchBytes_, farStream_);
		}
		primNow_ = false;
		{ // FixupState of the far boxes (after Reflux and AverageDownTo in the reference, src/simulation.hpp:1308-1312 — neither touches a far box)
			auto t = qkhost::traits<problem_t>();
			qkhost::check(qk_hydro_FixupState(specFar_.lev, farStream_, &t, densityFloor_, tempFloor_, useDualEnergy_, groupTableOf(specFar_, qkhost::tab(state_new_cc_[0])),
							  d_error_, d_fixFar_),
				      "qk_hydro_FixupState(far)");
		}
		QK_HOST_HIP(hipEventRecord(evFarDone_, farStream_));
		this->fillPhysicalBoundaries(state_new_cc_[0], QK_BOXES_LOCAL_ONLY);
		stage1LeftF1_ = !carryActive();
		if (afterAdvance_) {
			afterAdvance_(dt_lev); // incrementFluxRegisters, coarse side: the register cells lie in the near boxes
		}
		this->oldStateGhostsFilled_ = true;
	}
	// the verdict of advanceLevelBegin: both stages clean on every box, no error flag, no CFL violation
	auto advanceLevelJoin(double dt_lev) -> bool
	{
		this->activate();
		QK_HOST_HIP(hipStreamWaitEvent(qkhost::Runtime::get().computeStream(), evFarDone_, 0));
		StageWords w[2];
		readWords(w);
		if (w[0].err != 0 || w[1].err != 0) {
			amrex::Abort("density is negative in SyncDualEnergy! abort!!");
		}
		if (w[0].count != 0 || w[1].count != 0) {
			if (primHandoffChecked_ && primHandoffOk_) {
				++primHandoffDropped_;
				primBackoffLen_ = std::min(64, std::max(4, 2 * primBackoffLen_));
				primBackoff_ = primBackoffLen_;
			}
			return false;
		}
		primBackoffLen_ = 0;
		signal_[0] = w[1].sig[0];
		signal_[1] = w[1].sig[1];
		haveSignal_ = true;
		return !isCflViolated(dt_lev);
	}
	// FixupState of the near boxes after Reflux and AverageDownTo; with the far boxes' (advanceLevelBegin) the whole level has had it
	void fixupNear()
	{
		this->activate();
		if (d_fixSignal_ == nullptr) {
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_fixSignal_), 2 * sizeof(double)));
		}
		auto t = qkhost::traits<problem_t>();
		hipStream_t const cs = qkhost::Runtime::get().computeStream();
		qkhost::check(qk_hydro_FixupState(specNear_.lev, cs, &t, densityFloor_, tempFloor_, useDualEnergy_, groupTableOf(specNear_, qkhost::tab(state_new_cc_[0])), d_error_,
						  d_fixSignal_),
			      "qk_hydro_FixupState(near)");
		invalidateSignal();
		signalPending_ = true;
		fixFarPending_ = true; // (the far boxes' maxima wait in d_fixFar_: resolveSignal takes the larger)
	}
	// the speculative step did not stand: the states as they were before advanceLevelBegin (the old state was never written), the sticky error
	// words of the discarded attempt cleared
	void rollBackSpeculativeStep()
	{
		QK_HOST_HIP(hipStreamWaitEvent(qkhost::Runtime::get().computeStream(), evFarDone_, 0));
		std::swap(state_old_cc_[0], state_new_cc_[0]);
		QK_HOST_HIP(hipMemsetAsync(d_words_, 0, 8 * sizeof(int64_t), qkhost::Runtime::get().computeStream()));
		invalidateSignal();
		this->oldStateGhostsFilled_ = false;
		this->newStateGhostsFilled_ = false;
	}
	void clearErrorWords()
	{
		QK_HOST_HIP(hipMemsetAsync(d_words_, 0, 8 * sizeof(int64_t), qkhost::Runtime::get().computeStream()));
		invalidateSignal();
	}

	// advanceSingleTimestepAtLevel for a hydro level of a hierarchy: state_new <- advance(previous state_new) starting at `time`
	auto advanceLevel(double time, double dt_lev) -> bool
	{
		std::swap(state_old_cc_[0], state_new_cc_[0]);
		this->oldStateGhostsFilled_ = false;
		if constexpr (Physics_Traits<problem_t>::is_hydro_enabled) {
			if (!advanceHydroAtLevelWithRetries(time, dt_lev)) {
				return false;
			}
		} else { // reference src/QuokkaSimulation.hpp:681-685
			amrex::MultiFab::Copy(state_new_cc_[0], state_old_cc_[0], 0, 0, ncompHydro_, 0);
		}
		if constexpr (is_radiation_enabled_) { // :693
			subcycleRadiationAtLevel(time, dt_lev);
		}
		callAfterLevelAdvance(time, dt_lev);
		return true;
	}

      private:
	void construct()
	{
		amrex::ParmParse hpp("hydro"); // reference src/QuokkaSimulation.hpp:340-350
		hpp.query("rk_integrator_order", integratorOrder_);
		hpp.query("reconstruction_order", reconstructionOrder_);
		hpp.query("use_dual_energy", useDualEnergy_);
		hpp.query("abort_on_fofc_failure", abortOnFofcFailure_);
		hpp.query("artificial_viscosity_coefficient", artificialViscosityK_);
		{
			amrex::ParmParse qpp("qk");
			qpp.query("fused_fofc", fusedFofc_);
		}
		hpp.query("rk2_carry_rhs", rk2CarryRhs_); // extension of this host: the carried-rhs form of the RK2 average (<= 1e-12; quokka_amd.h)
		{
			// cooling.* (reference src/QuokkaSimulation.hpp:352-376): the Strang-split source from tabulated cooling curves.  The Cloudy tables of the
			// cloudy_cooling_tools are read by the library's own reader of the HDF5 format; Grackle's table files are not in the reference tree
			// (extern/grackle_data_files is an empty submodule) and that table type is refused.
			amrex::ParmParse cpp("cooling");
			int alwaysReadTables = 0;
			cpp.query("enabled", enableCooling_);
			cpp.query("read_tables_even_if_disabled", alwaysReadTables);
			cpp.query("cooling_table_type", coolingTableType_);
			cpp.query("hdf5_data_file", coolingTableFilename_);
			if ((enableCooling_ == 1) || (alwaysReadTables == 1)) {
				if (coolingTableType_ == "cloudy_cooling_tools") {
					amrex::Print() << "Reading cloudy-cooling-tools tables...\n";
					quokka::TabulatedCooling::readCloudyData(coolingTableFilename_, cloudyTables_);
				} else if (coolingTableType_ == "grackle") {
					amrex::Print() << "Reading Grackle tables...\n";
					quokka::GrackleLikeCooling::readGrackleData(coolingTableFilename_, grackleTables_); // (refuses: not built)
				} else {
					amrex::Abort("Invalid cooling table type!");
				}
			}
		}
		amrex::ParmParse rpp("radiation"); // reference src/QuokkaSimulation.hpp:353-358
		rpp.query("reconstruction_order", radiationReconstructionOrder_);
		rpp.query("cfl", radiationCflNumber_);
		rpp.query("dust_gas_interaction_coeff", dustGasInteractionCoeff_);
		rpp.query("max_substeps", maxSubsteps_);
		{
			int ti = 0;
			rpp.query("source_is_time_independent", ti);
			radSourceTimeIndependent_ = (ti != 0);
		}
		std::string walltime;
		if (amrex::ParmParse().query("max_walltime", walltime)) { // H:M:S (reference src/simulation.hpp:618-628)
			int h = 0, m = 0, sec = 0;
			if (std::sscanf(walltime.c_str(), "%d:%d:%d", &h, &m, &sec) == 3) {
				maxWalltime_ = 3600L * h + 60L * m + sec;
			}
		}
		defineComponentNames();
		allocate();
	}

      public:
	// reference src/QuokkaSimulation.hpp:283-310 (no passive scalars in this build)
	void defineComponentNames()
	{
		this->componentNames_cc_ = {"gasDensity", "x-GasMomentum", "y-GasMomentum", "z-GasMomentum", "gasEnergy", "gasInternalEnergy"};
		if constexpr (is_radiation_enabled_) {
			for (int i = 0; i < Physics_Traits<problem_t>::nGroups; ++i) {
				for (auto const *name : {"radEnergy-Group", "x-RadFlux-Group", "y-RadFlux-Group", "z-RadFlux-Group"}) {
					this->componentNames_cc_.push_back(name + std::to_string(i));
				}
			}
		}
	}

	// on-disk formats (quokka_io.hpp); all levels of the hierarchy, defined in quokka_amr.hpp
	void WritePlotFile();	    // reference src/simulation.hpp:2294-2336
	void WriteCheckpointFile(); // reference src/simulation.hpp:2564-2666
	long maxWalltime_ = 0;	    // seconds, 0: no limit
	int lastPlotFileStep_ = 0, lastChkFileStep_ = 0;
	// output schedule of AMRSimulation::setInitialConditions / evolve (reference src/simulation.hpp:657-684, 910-941, 983-1003)
	void outputAfterInitialConditions()
	{
		if (this->restart_chkfile.empty() && this->checkpointInterval_ > 0) {
			WriteCheckpointFile();
		}
		if (this->plotfileInterval_ > 0) {
			WritePlotFile();
		}
		lastPlotFileStep_ = lastChkFileStep_ = istep[0];
	}
	void outputAfterStep(int step)
	{
		if (this->plotfileInterval_ > 0 && (step + 1) % this->plotfileInterval_ == 0) {
			lastPlotFileStep_ = step + 1;
			WritePlotFile();
		}
		if (this->checkpointInterval_ > 0 && (step + 1) % this->checkpointInterval_ == 0) { // after the plotfile, like the reference
			lastChkFileStep_ = step + 1;
			WriteCheckpointFile();
		}
	}
	void outputAfterEvolve()
	{
		if (this->plotfileInterval_ > 0 && istep[0] > lastPlotFileStep_) {
			WritePlotFile();
		}
		if (this->checkpointInterval_ > 0 && istep[0] > lastChkFileStep_) {
			WriteCheckpointFile();
		}
	}
	[[nodiscard]] auto walltimeExceeded(std::chrono::steady_clock::time_point t0) const -> bool
	{
		return maxWalltime_ > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.9 * static_cast<double>(maxWalltime_);
	}

	void setInitialConditionsOnGrid(quokka::grid const &grid_elem) override;
	void setInitialConditionsOnGridFaceVars(quokka::grid const &grid_elem) override; // (a problem with a face-centred state specialises it)
	using AMRSimulation<problem_t>::componentNames_fc;
	void preCalculateInitialConditions() override;
	void computeAfterEvolve(amrex::Vector<amrex::Real> &initSumCons) override;
	void computeAfterTimestep(); // reference src/simulation.hpp:228, :890 (default: nothing)
	// output hooks (reference src/QuokkaSimulation.hpp:190-196,550-566): derived plot variables, projections, statistics — defaults do nothing
	void ComputeDerivedVar(int lev, std::string const &dname, amrex::MultiFab &mf, int ncomp) const;
	[[nodiscard]] auto ComputeProjections(int dir) const -> std::unordered_map<std::string, amrex::BaseFab<amrex::Real>>;
	auto ComputeStatistics() -> std::map<std::string, amrex::Real>;
	// the mean of user_f(i, j, k, state) over the planes normal to `axis` (QuokkaSimulation.hpp:843-881): evaluated on every level, averaged
	// down, summed on level 0.  Defined in quokka_amr.hpp.
	template <typename F> auto computeAxisAlignedProfile(int axis, F const &user_f) -> amrex::Gpu::HostVector<amrex::Real>;
	// Strang-split source terms a problem may add (QuokkaSimulation.hpp:235): called with dt/2 on the old state before the hydro update and on
	// the new state after it (:1048, :1318)
	void addStrangSplitSources(amrex::MultiFab &state, int lev, amrex::Real time, amrex::Real dt_lev);
	void createInitialParticles();
	void computeBeforeTimestep();
	// the two user hooks as the drivers call them: a specialised hook may change the state, so the signal speeds cached by the last stage are dropped
	void dropCachedSignal() { invalidateSignal(); }
	void callAfterTimestep()
	{
		computeAfterTimestep();
		if (!afterTimestepIsDefault_) {
			invalidateSignal();
		}
	}
	// operator-split work of the problem after a level has advanced (reference src/QuokkaSimulation.hpp:184,508,699-700; default: nothing)
	void computeAfterLevelAdvance(int lev, amrex::Real time, amrex::Real dt_lev, int ncycle);
	void callAfterLevelAdvance(double time, double dt_lev)
	{
		computeAfterLevelAdvance(this->amrLevel_, time, dt_lev, 1);
		if (!afterLevelAdvanceIsDefault_) {
			invalidateSignal();
		}
	}
	void callBeforeTimestep()
	{
		computeBeforeTimestep();
		if (!beforeTimestepIsDefault_) {
			invalidateSignal();
		}
	}
	void computeReferenceSolution(amrex::MultiFab & /*ref*/, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const & /*dx*/,
				      amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> const & /*prob_lo*/)
	{
	}

	// ------------------------------------------------------------------ dt (reference src/simulation.hpp:703-818)
	void computeTimestep()
	{
		double m = 0.0;
		if constexpr (!Physics_Traits<problem_t>::is_hydro_enabled) { // radiation only (reference src/QuokkaSimulation.hpp:421-424): c_hat in every cell
			static_assert(is_radiation_enabled_, "At least one of hydro or radiation must be enabled! Cannot compute a time step.");
			m = RadSystem<problem_t>::c_hat_;
		} else {
			resolveSignal();
			m = (haveSignal_ ? signal_[1] : HydroSystem<problem_t>::maxSignalSpeedLocal(state_new_cc_[0], 1));
		}
		if constexpr (is_radiation_enabled_ && Physics_Traits<problem_t>::is_hydro_enabled) {
			// reference src/QuokkaSimulation.hpp:421-434: per cell max(c_hat / maxSubsteps, hydro signal); the max over cells commutes
			m = std::max(RadSystem<problem_t>::c_hat_ / static_cast<double>(maxSubsteps_), m);
		}
		double dt_tmp = cflNumber_ * (minDx() / m);
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

	// the conservation report at the end of evolve (reference src/simulation.hpp:959-970): level 0 holds the average of every finer level, so
	// its sum is the composite integral
	void printConservation(amrex::Vector<amrex::Real> const &init_sum_cons, double vol)
	{
		int const nc = Physics_Indices<problem_t>::nvarTotal_cc;
		for (int n = 0; n < nc && n < static_cast<int>(init_sum_cons.size()); ++n) {
			amrex::Real const final_sum = state_new_cc_[0].sum(n) * vol;
			amrex::Real const abs_err = (final_sum - init_sum_cons[n]);
			std::string const name = n < static_cast<int>(this->componentNames_cc_.size()) ? this->componentNames_cc_[n] : ("component" + std::to_string(n));
			amrex::Print() << "Initial " << name << " = " << init_sum_cons[n] << "\n";
			amrex::Print() << "\tabsolute conservation error = " << abs_err << "\n";
			if (init_sum_cons[n] != 0.0) {
				amrex::Print() << "\trelative conservation error = " << abs_err / init_sum_cons[n] << "\n";
			}
			amrex::Print() << "\n";
		}
	}

	// amr.max_level > 0: the level machinery of quokka_amr.hpp takes over (this object is level 0); defined there
	void setInitialConditions();
	void evolve();
	std::shared_ptr<AmrDriver<problem_t, QuokkaSimulation<problem_t>>> amr_;

	// ------------------------------------------------------------------ evolve (reference src/simulation.hpp:827-981)
	void evolveSingleLevel()
	{
		AMREX_ALWAYS_ASSERT(this->areInitialConditionsDefined_);
		int const nc = Physics_Indices<problem_t>::nvarTotal_cc;
		double const vol = AMREX_D_TERM(geom[0].dx[0], *geom[0].dx[1], *geom[0].dx[2]);
		amrex::Vector<amrex::Real> init_sum_cons(nc);
		for (int n = 0; n < nc; ++n) {
			init_sum_cons[n] = state_new_cc_[0].sum(n) * vol;
		}
		if constexpr (is_radiation_enabled_) {
			fillRadEnergySource(tNew_[0]); // host-evaluated hook: a time-independent source is set up before the clock starts
		}
		QK_HOST_HIP(hipDeviceSynchronize());
		auto const t0 = std::chrono::steady_clock::now();
		double cur_time = tNew_[0];
		for (int step = istep[0]; step < maxTimesteps_ && cur_time < stopTime_; ++step) {
			computeTimestep();
			callBeforeTimestep(); // reference src/simulation.hpp:864-867: after computeTimestep
			double const time = tNew_[0];
			tNew_[0] += dt_[0];
			std::swap(state_old_cc_[0], state_new_cc_[0]);
			if constexpr (Physics_Traits<problem_t>::is_hydro_enabled) {
				if (!advanceHydroAtLevelWithRetries(time, dt_[0])) {
					amrex::Abort("QUOKKA FATAL ERROR: Hydro update exceeded max_retries on level 0");
				}
			} else { // copy hydro vars from state_old_cc_ to state_new_cc_ (reference src/QuokkaSimulation.hpp:681-685)
				amrex::MultiFab::Copy(state_new_cc_[0], state_old_cc_[0], 0, 0, ncompHydro_, 0);
			}
			if constexpr (is_radiation_enabled_) { // advanceSingleTimestepAtLevel (reference src/QuokkaSimulation.hpp:653-707)
				subcycleRadiationAtLevel(time, dt_[0]);
			}
			callAfterLevelAdvance(time, dt_[0]);
			++istep[0];
			this->cellUpdates_ += this->CountCells(0);
			cur_time += dt_[0];
			tNew_[0] = cur_time;
			callAfterTimestep(); // reference src/simulation.hpp:890
			outputAfterStep(step);
			if (cur_time >= stopTime_ - 1.e-6 * dt_[0]) {
				break;
			}
			if (walltimeExceeded(t0)) {
				break;
			}
		}
		QK_HOST_HIP(hipDeviceSynchronize());
		elapsedSeconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		outputAfterEvolve();
		this->computeAfterEvolve(init_sum_cons);
		printConservation(init_sum_cons, AMREX_D_TERM(geom[0].dx[0], *geom[0].dx[1], *geom[0].dx[2]));
		double const microseconds_per_update = 1.0e6 * elapsedSeconds_ / static_cast<double>(this->cellUpdates_);
		amrex::Print() << "Performance figure-of-merit: " << microseconds_per_update << " μs/zone-update [" << 1.0 / microseconds_per_update
			       << " Mupdates/s]\n";
		// (not in the reference's output: how often the first-order flux correction and the retry loop ran — bench.py's full_run block reads it)
		amrex::Print() << "qk counters: steps=" << istep[0] << " fofc_stages=" << fofcStages_ << " retries=" << retries_ << " elapsed_s=" << elapsedSeconds_
			       << " sim_time=" << tNew_[0] << " prim_handoff=" << ((primHandoffChecked_ && primHandoffOk_) ? 1 : 0)
			       << " prim_handoff_dropped=" << primHandoffDropped_ << "\n";
	}

	// ------------------------------------------------------------------ hydro advance (reference src/QuokkaSimulation.hpp:885-1322)
	auto advanceHydroAtLevelWithRetries(double time, double dt_lev) -> bool
	{
		this->activate();
		const int max_retries = 6;
		bool success = false;
		for (int retry_count = 0; retry_count <= max_retries; ++retry_count) {
			const int nsubsteps = 1 << retry_count;
			const double dt_step = dt_lev / nsubsteps;
			if (retry_count > 0) {
				++retries_;
			}
			if (beforeAttempt_) {
				beforeAttempt_(retry_count); // reference src/QuokkaSimulation.hpp:894-900 (save), :919-929 (reset / restore)
			}
			bool const direct = strangSourcesAreDefault_ && enableCooling_ == 0 && nsubsteps == 1;
			if (!direct) {
				amrex::MultiFab::Copy(state_old_tmp_, state_old_cc_[0]);
			}
			for (int substep = 0; substep < nsubsteps; ++substep) {
				if (substep > 0) {
					amrex::MultiFab::Copy(state_old_tmp_, state_new_cc_[0]);
				}
				success = advanceHydroAtLevel(direct ? state_old_cc_[0] : state_old_tmp_, time + substep * dt_step, dt_step);
				if (!success) {
					break;
				}
			}
			if (success) {
				// the stage-1 fill of a direct attempt was applied to state_old_cc_ itself: its ghost cells hold the fill at `time`, and nothing
				// writes the old state afterwards (hydro only: the radiation subcycle mirrors its substeps into it)
				this->oldStateGhostsFilled_ = direct && !is_radiation_enabled_;
				break;
			}
		}
		return success;
	}

	// reference src/QuokkaSimulation.hpp:519-547: the built-in cooling source, then the problem's own; false = the cooling integrator failed
	auto addStrangSplitSourcesWithBuiltin(amrex::MultiFab &state, int lev, amrex::Real time, amrex::Real dt) -> bool
	{
		bool cool_success = true;
		if (enableCooling_ == 1) {
			if (coolingTableType_ == "cloudy_cooling_tools") {
				cool_success = quokka::TabulatedCooling::computeCooling<problem_t>(state, dt, cloudyTables_, tempFloor_);
			} else {
				amrex::Abort("Invalid cooling table type!");
			}
		}
		addStrangSplitSources(state, lev, time, dt);
		return cool_success;
	}

	auto advanceHydroAtLevel(amrex::MultiFab &state_old_cc_tmp, double time, double dt_lev) -> bool
	{
		invalidateSignal();
		// first half of the Strang-split source terms, on the (temporary) old state (reference src/QuokkaSimulation.hpp:1048-1054)
		if (!addStrangSplitSourcesWithBuiltin(state_old_cc_tmp, 0, time, 0.5 * dt_lev)) {
			return false;
		}
		int pair = -1;
		if constexpr (fusedEligible()) {
			if (integratorOrder_ == 2 && speculateStage2_ != 0) {
				pair = stagePairSpeculative(state_old_cc_tmp, time, dt_lev);
				if (pair == 0) {
					return false;
				}
			}
		}
		fillTime_ = time; // reference src/QuokkaSimulation.hpp:1076 (stage 1), :1204 (stage 2: time + dt_lev)
		if (pair == 1) {
			// (both stages done)
		} else if (!fillAndStage(1, state_old_cc_tmp, state_old_cc_tmp, state_inter_cc_, dt_lev)) {
			return false;
		} else if (integratorOrder_ == 2) {
			fillTime_ = time + dt_lev;
			if (!fillAndStage(2, state_inter_cc_, state_old_cc_tmp, state_new_cc_[0], dt_lev)) {
				return false;
			}
		} else {
			amrex::MultiFab::Copy(state_new_cc_[0], state_inter_cc_);
		}
		if (unfusedRan_) { // (a fused stage reports its error flag with its redo count: fusedEnd)
			unfusedRan_ = false;
			int err = 0;
			QK_HOST_HIP(hipMemcpy(&err, d_error_, sizeof(int), hipMemcpyDeviceToHost));
			err = qkhost::Comm::get().allReduceMax(err);
			if (err != 0) {
				amrex::Abort("density is negative in SyncDualEnergy! abort!!");
			}
		}
		// second half of the Strang-split sources on the new state, THEN the CFL check on what they left (reference :1318-1321: the sources run
		// unconditionally and `return !isCflViolated(...) && burn_success` sees the post-source state — a heating source can raise the signal
		// speed past the limit).  Default sources that change nothing keep the signal the final stage reduced.
		bool ok = true;
		if (!strangSourcesAreDefault_ || enableCooling_ == 1) {
			ok = addStrangSplitSourcesWithBuiltin(state_new_cc_[0], 0, time + dt_lev, 0.5 * dt_lev);
			invalidateSignal();
			ok = !isCflViolated(dt_lev) && ok;
		} else {
			ok = !isCflViolated(dt_lev);
			if (ok) {
				ok = addStrangSplitSourcesWithBuiltin(state_new_cc_[0], 0, time + dt_lev, 0.5 * dt_lev);
			}
		}
		if (ok && afterAdvance_) {
			afterAdvance_(dt_lev); // incrementFluxRegisters (reference src/QuokkaSimulation.hpp:1303-1306)
		}
		return ok;
	}

	auto isCflViolated(double dt_actual) -> bool // reference src/QuokkaSimulation.hpp:992-1013
	{
		resolveSignal();
		double const max_signal = haveSignal_ ? signal_[0] : HydroSystem<problem_t>::maxSignalSpeedLocal(state_new_cc_[0], 0);
		double const dt_cfl = cflNumber_ * (minDx() / max_signal);
		return dt_actual > (1.1 * dt_cfl);
	}

	// ------------------------------------------------------------------ radiation subcycle (reference src/QuokkaSimulation.hpp:397-406,1576-1882)
	[[nodiscard]] auto computeNumberOfRadiationSubsteps(double dt_lev_hydro) const -> int
	{
		double const dtrad_tmp = radiationCflNumber_ * (minDx() / RadSystem<problem_t>::c_hat_);
		return static_cast<int>(std::ceil(dt_lev_hydro / dtrad_tmp));
	}

	// operatorSplitSourceTerms (:1859-1882).  The problem's SetRadEnergySource launches its own kernel on the source array before every
	// source-term call, as the reference does (:1866-1873).  A deck may declare the source time-independent
	// (`radiation.source_is_time_independent = 1`, an extension of this host; default 0): it is then evaluated once — the kernel was 9 % of a
	// RadhydroShell step (160 launches per step).  Nothing is inferred: a source that is switched off at some time (RadSuOlson) needs the default.
	void fillRadEnergySource(double time)
	{
		if (radSourceTimeIndependent_ && radSourceFilled_) {
			return;
		}
		// The hook is a function of (box, dx, prob_lo, prob_hi, time): a call with the time of the last evaluation finds its values in the array.
		// Both stages of a substep pass time_subcycle + dt_radiation (reference :1638, :1656 -> :1872), so the second is free — nothing is assumed
		// about how the source depends on time.
		if (radSourceFilled_ && time == radSourceTime_) {
			return;
		}
		auto const &g = geom[0];
		{
			amrex::qk_parfor_batch_scope batch; // the hook's one-box ParallelFor calls leave as one launch over all boxes (amrex_mini.hpp)
			for (int b = 0; b < radEnergySource_.size(); ++b) {
				auto arr = radEnergySource_.array(b);
				RadSystem<problem_t>::SetRadEnergySource(arr, radEnergySource_.validbox(b), g.CellSizeArray(), g.ProbLoArray(), g.ProbHiArray(), time);
			}
		}
		radSourceFilled_ = true;
		radSourceTime_ = time;
	}

	void operatorSplitSourceTerms(double time, double dt, int stage, bool mirror = false)
	{
		fillRadEnergySource(time + dt);
		// debugging aid: QK_DUMP_BEFORE_SOURCE="<level> <tmin> <prefix>" writes what the first source-term launch of that level at time >= tmin is about to read —
		// every fab of the state with its ghost cells, then the source array — to <prefix>.rank<r>.bin (doubles; boxes in <prefix>.rank<r>.txt)
		if (char const *e = std::getenv("QK_DUMP_BEFORE_SOURCE")) {
			int lev = -1;
			double tmin = 0;
			char prefix[512];
			static bool dumped = false;
			if (!dumped && std::sscanf(e, "%d %lf %500s", &lev, &tmin, prefix) == 3 && lev == this->amrLevel_ && time >= tmin) {
				dumped = true;
				std::ostringstream head;
				head << "time " << std::setprecision(17) << time << " dt " << dt << " stage " << stage;
				qkhost::dumpFabs(prefix, head.str(), {&state_new_cc_[0], &radEnergySource_});
			}
		}
		if constexpr (Physics_Traits<problem_t>::nGroups <= 1) { // :1875-1881
			RadSystem<problem_t>::AddSourceTermsSingleGroup(state_new_cc_[0], radEnergySource_, dt, stage, d_radCounter_ + 4 * radCounterSlot_, d_radFailure_,
									mirror ? &state_old_cc_[0] : nullptr);
		} else {
			RadSystem<problem_t>::AddSourceTermsMultiGroup(state_new_cc_[0], radEnergySource_, dt, stage, d_radCounter_ + 4 * radCounterSlot_, d_radFailure_);
		}
	}

	// use_wavespeed_correction_ (:133, :1958-1960): the factors of ComputeFluxes<DIR>'s optional correction for the state whose fluxes are about to be
	// taken (its ghost cells filled in EVERY component: ComputeCellOpticalDepth reads the gas either side of a face); nullptr when off
	std::array<amrex::MultiFab, AMREX_SPACEDIM> radEps_;
	bool radEpsDefined_ = false;
	auto wavespeedEps(amrex::MultiFab const &state) -> std::array<amrex::MultiFab, AMREX_SPACEDIM> const *
	{
		if (!use_wavespeed_correction_) {
			return nullptr;
		}
		if (!radEpsDefined_) {
			radEpsDefined_ = true;
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				radEps_[d].define(grids_, Physics_Traits<problem_t>::nGroups > 1 ? Physics_Traits<problem_t>::nGroups : 1, 0, d);
			}
		}
		RadSystem<problem_t>::ComputeWavespeedCorrection(state, geom[0].CellSizeArray(), radEps_);
		return &radEps_;
	}
	void fillGhostsForTransport(amrex::MultiFab &state)
	{
		if (use_wavespeed_correction_) {
			this->fillBoundaryConditions(state);
		} else {
			this->fillRadiationGhosts(state);
		}
	}

	void advanceRadiationForwardEuler(double dt_radiation) // :1790-1821
	{
		fillTime_ = radTime_; // (a refined level: its ghost cells come from the parent at the time of the substep, :1743)
		fillGhostsForTransport(state_old_cc_[0]);
		if (radFusedActive()) {
			RadSystem<problem_t>::stageFused(1, radiationReconstructionOrder_, state_old_cc_[0], state_old_cc_[0], state_new_cc_[0], radAcc_,
							 afterRadStage_ ? &radFluxOld_ : nullptr, dt_radiation, geom[0].CellSizeArray(), wavespeedEps(state_old_cc_[0]));
		} else {
			RadSystem<problem_t>::computeRadiationFluxes(state_old_cc_[0], radFluxOld_, radiationReconstructionOrder_, wavespeedEps(state_old_cc_[0]));
			RadSystem<problem_t>::PredictStep(state_old_cc_[0], state_new_cc_[0], radFluxOld_, dt_radiation, geom[0].CellSizeArray());
		}
		if (afterRadStage_) {
			afterRadStage_(radFluxOld_, dt_radiation); // incrementFluxRegisters(..., 0.5 * dt_radiation) (:1818)
		}
	}

	void advanceRadiationMidpointRK2(double dt_radiation) // :1823-1857 (the fluxes of the old state are reused, not recomputed)
	{
		fillTime_ = radTime_ + dt_radiation; // :1764
		fillGhostsForTransport(state_new_cc_[0]);
		if (radFusedActive()) { // (the Z sweep writes state_new in place: it marches every column in one thread and reads no other column)
			RadSystem<problem_t>::stageFused(2, radiationReconstructionOrder_, state_new_cc_[0], state_old_cc_[0], state_new_cc_[0], radAcc_,
							 afterRadStage_ ? &radFlux_ : nullptr, dt_radiation, geom[0].CellSizeArray(), wavespeedEps(state_new_cc_[0]));
		} else {
			RadSystem<problem_t>::computeRadiationFluxes(state_new_cc_[0], radFlux_, radiationReconstructionOrder_, wavespeedEps(state_new_cc_[0]));
			RadSystem<problem_t>::AddFluxesRK2(state_new_cc_[0], state_old_cc_[0], state_new_cc_[0], radFluxOld_, radFlux_, dt_radiation, geom[0].CellSizeArray());
		}
		if (afterRadStage_) {
			afterRadStage_(radFlux_, dt_radiation); // :1854
		}
	}

	// qk_rad_stage_fused serves 3-D builds (one set of sweeps per photon group; deck: qk.fused_radiation = 0 keeps the separate operators; tests)
	amrex::MultiFab radAcc_;
	int radFused_ = -1;
	auto radFusedActive() -> bool
	{
		if (radFused_ < 0) {
			radFused_ = 0;
			if constexpr (AMREX_SPACEDIM == 3) {
				radFused_ = 1;
				amrex::ParmParse("qk").query("fused_radiation", radFused_);
				if (radFused_ != 0) {
					radAcc_.define(grids_, RadSystem<problem_t>::nvarHyperbolic_, 0);
				}
			}
		}
		return radFused_ == 1;
	}

	bool radMirror_ = [] {
		int v = 1;
		amrex::ParmParse("qk").query("rad_mirror", v);
		return v != 0;
	}();

	void subcycleRadiationAtLevel(double time, double dt_lev_hydro)
	{
		int nsubSteps = 1;
		double dt_radiation = dt_lev_hydro;
		if (Physics_Traits<problem_t>::is_hydro_enabled && !(this->constantDt_ > 0.0)) { // reference src/QuokkaSimulation.hpp:1583
			nsubSteps = computeNumberOfRadiationSubsteps(dt_lev_hydro);
			dt_radiation = dt_lev_hydro / static_cast<double>(nsubSteps);
		}
		if (!(nsubSteps >= 1 && nsubSteps <= maxSubsteps_ + 1 && dt_radiation > 0.0)) {
			amrex::Abort("radiation substep assertion failed (reference src/QuokkaSimulation.hpp:1596-1598)");
		}
		invalidateSignal(); // the source terms change the gas state
		double time_subcycle = time;
		int const r0 = RadSystem<problem_t>::nstartHyperbolic_;
		// Newton counters: one slot of 4 ints per substep (a slot stays below 2^31 at any box size); one host read per level advance
		if (nsubSteps > radCounterSlots_) {
			(void)hipFree(d_radCounter_);
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_radCounter_), 4 * sizeof(int) * static_cast<size_t>(nsubSteps)));
			radCounterSlots_ = nsubSteps;
		}
		QK_HOST_HIP(hipMemsetAsync(d_radCounter_, 0, 4 * sizeof(int) * static_cast<size_t>(nsubSteps), qkhost::Runtime::get().computeStream()));
		QK_HOST_HIP(hipMemsetAsync(d_radFailure_, 0, 3 * sizeof(int), qkhost::Runtime::get().computeStream()));
		bool mirrored = false; // swapRadiationState already done by the source-term kernel of the substep before (`qk.rad_mirror`, default 1)
		for (int i = 0; i < nsubSteps; ++i) {
			if (i > 0 && !mirrored) { // swapRadiationState (:1783-1788)
				amrex::MultiFab::Copy(state_old_cc_[0], state_new_cc_[0], r0, r0, RadSystem<problem_t>::nvarHyperbolic_, 0);
			}
			radCounterSlot_ = i;
			radTime_ = time_subcycle;
			advanceRadiationForwardEuler(dt_radiation);
			operatorSplitSourceTerms(time_subcycle, dt_radiation, 1); // IMEX_a22 > 0
			advanceRadiationMidpointRK2(dt_radiation);
			mirrored = radMirror_ && Physics_Traits<problem_t>::nGroups <= 1 && i < nsubSteps - 1;
			operatorSplitSourceTerms(time_subcycle, dt_radiation, 2, mirrored);
			time_subcycle += dt_radiation;
			radiationCellUpdates_ += this->CountCells(0);
		}
		std::vector<int> cnt(4 * static_cast<size_t>(nsubSteps));
		int fail[3];
		QK_HOST_HIP(hipMemcpy(cnt.data(), d_radCounter_, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
		QK_HOST_HIP(hipMemcpy(fail, d_radFailure_, sizeof(fail), hipMemcpyDeviceToHost));
		double v[5] = {0, 0, double(fail[0]), double(fail[1]), double(fail[2])};
		int maxIt = 0;
		for (int i = 0; i < nsubSteps; ++i) {
			v[0] += cnt[4 * i];
			v[1] += cnt[4 * i + 1];
			maxIt = std::max(maxIt, cnt[4 * i + 2]);
		}
		if (fail[0] + fail[1] + fail[2] > 0 && std::getenv("QK_AMR_VERBOSE") != nullptr) { // (which level, which rank: before the counts are reduced)
			std::fprintf(stderr, "rank %d level %d (%d boxes here): radiation source term failures %d %d %d in %d substeps at t = %.17g\n", qkhost::Comm::get().rank,
				     this->amrLevel_, static_cast<int>(this->grids_.size()), fail[0], fail[1], fail[2], nsubSteps, time_subcycle);
		}
		if (qkhost::Comm::get().size > 1) { // counters over all ranks (the reference reduces them when it prints them, QuokkaSimulation.hpp:1690-1720)
			qkhost::Comm::get().allReduce(v, 5, qkhost::Comm::Op::sum);
			maxIt = qkhost::Comm::get().allReduceMax(maxIt);
		}
		radSolves_ += static_cast<int64_t>(v[0]);
		radNewtonIterations_ += static_cast<int64_t>(v[1]);
		radMaxNewtonIterations_ = std::max(radMaxNewtonIterations_, maxIt);
		if (v[3] > 0) {
			amrex::Abort("Newton-Raphson iteration for dust temperature failed to converge or dust temperature is negative!");
		}
		if (v[2] > 0 && std::getenv("QK_IGNORE_RAD_FAILURE") == nullptr) { // (the variable: a debugging aid — run on, to look at the state that failed)
			amrex::Abort("Newton-Raphson iteration for matter-radiation coupling failed to converge!");
		}
		if (v[4] > 0 && std::getenv("QK_IGNORE_RAD_FAILURE") == nullptr) {
			amrex::Abort("Outer iteration for matter-radiation coupling failed to converge!");
		}
	}

      private:
	amrex::MultiFab state_old_tmp_, state_inter_cc_, primVar_, rhs_;
	std::array<amrex::MultiFab, 3> flatCoefs_;
	std::array<amrex::MultiFab, AMREX_SPACEDIM> halfFlux_, halfVel_, flux_, vel_, FOflux_, FOvel_, rk2flux_, rk2vel_, leftState_, rightState_;
	amrex::iMultiFab redoFlag_;
	qk_ghost_plan *flagPlan_ = nullptr;
	// The words a fused stage reports in — [max signal, max signal for dt] (double), [redo count] (int64), [error flag of SyncDualEnergy] (int) —
	// in TWO slots of one allocation: an RK2 step enqueues both stages before it reads either (stage 1 reports in slot 0, stage 2 in slot 1:
	// one device -> host copy and one host synchronisation per step instead of four); everything else uses slot 0.
	int64_t *d_words_ = nullptr;
	int64_t *d_count_ = nullptr; // = slot 0
	int *d_error_ = nullptr;
	double *d_signal_ = nullptr;
	struct StageWords {
		double sig[2];
		int64_t count;
		int64_t err;
	};
	int speculateStage2_ = 1; // deck: qk.speculate_stage2 (0: the stages are read back one by one; tests)
	void *scratch_ = nullptr;
	int64_t scratchBytes_ = 0;
	double signal_[2] = {0, 0};
	bool haveSignal_ = false;
	bool signalPending_ = false; // d_fixSignal_ holds the maxima of the current state_new_cc_
	double *d_fixSignal_ = nullptr;
	// Set by the DEFAULT addStrangSplitSources (which does nothing else), known after the first call.  Without a specialised hook the advance
	// needs no private copy of the old state (nothing modifies it: the stages read U_old and write elsewhere) and the signal speeds of the
	// final stage's epilogue stay valid for the next computeTimestep: 0.4 ms (copy) + 0.4 ms (k_maxSignal) per Sedov 256^3 step.
	bool strangSourcesAreDefault_ = false;
	bool afterLevelAdvanceIsDefault_ = false;
	std::array<amrex::MultiFab, AMREX_SPACEDIM> radFluxOld_, radFlux_;
	amrex::MultiFab radEnergySource_;
	int *d_radCounter_ = nullptr, *d_radFailure_ = nullptr;
	int radCounterSlots_ = 1, radCounterSlot_ = 0;
	bool radSourceFilled_ = false;
	double radSourceTime_ = 0.0; // the time argument of the last SetRadEnergySource evaluation (fillRadEnergySource)

	[[nodiscard]] auto minDx() const -> double
	{
		double m = geom[0].dx[0];
		for (int d = 1; d < AMREX_SPACEDIM; ++d) {
			m = std::min(m, geom[0].dx[d]);
		}
		return m;
	}

	void allocate()
	{
		int const nc = Physics_Indices<problem_t>::nvarTotal_cc;
		state_old_tmp_.define(grids_, nc, nghost_cc_);
		state_inter_cc_.define(grids_, nc, nghost_cc_);
		state_inter_cc_.setVal(0);
		primVar_.define(grids_, ncompHydro_, nghost_cc_);
		rhs_.define(grids_, ncompHydro_, 0);
		redoFlag_.define(grids_, 1, 1);
		redoFlag_.setVal(0);
		for (int d = 0; d < 3; ++d) {
			flatCoefs_[d].define(grids_, 1, 2);
			flatCoefs_[d].setVal(1.0);
		}
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			for (auto *f : {&halfFlux_[d], &flux_[d], &FOflux_[d], &rk2flux_[d]}) {
				f->define(grids_, ncompHydro_, 0, d);
			}
			for (auto *v : {&halfVel_[d], &vel_[d], &FOvel_[d], &rk2vel_[d]}) {
				v->define(grids_, 1, 0, d);
			}
			leftState_[d].define(grids_, ncompHydro_, 1, d);
			rightState_[d].define(grids_, ncompHydro_, 1, d);
		}
		if constexpr (is_radiation_enabled_) {
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				radFluxOld_[d].define(grids_, RadSystem<problem_t>::nvarHyperbolic_, 0, d);
				radFlux_[d].define(grids_, RadSystem<problem_t>::nvarHyperbolic_, 0, d);
			}
			radEnergySource_.define(grids_, Physics_Traits<problem_t>::nGroups, 0); // :1866-1869
			radEnergySource_.setVal(0);
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_radCounter_), 4 * sizeof(int)));
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_radFailure_), 3 * sizeof(int)));
		}
		QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_words_), 8 * sizeof(int64_t)));
		QK_HOST_HIP(hipMemsetAsync(d_words_, 0, 8 * sizeof(int64_t), nullptr));
		d_signal_ = reinterpret_cast<double *>(d_words_);
		d_count_ = d_words_ + 2;
		d_error_ = reinterpret_cast<int *>(d_words_ + 3);
		{
			amrex::ParmParse pq("qk");
			pq.query("speculate_stage2", speculateStage2_);
		}
		auto t = qkhost::traits<problem_t>();
		if constexpr (HydroSystem<problem_t>::nscalars_ <= 3 && Physics_Traits<problem_t>::numMassScalars == 0 && Physics_Traits<problem_t>::is_hydro_enabled) {
			scratchBytes_ = qk_hydro_stage_scratch_bytes(qkhost::Runtime::get().lev, &t);
			QK_HOST_HIP(hipMalloc(&scratch_, std::max<size_t>(static_cast<size_t>(scratchBytes_), 8))); // (never null: a level may be empty on this rank)
		}
		// redoFlag.FillBoundary plan (1 ghost, 1 comp)
		auto const &g = geom[0];
		qk_geometry qg{};
		std::vector<qk_box> qb;
		for (auto const &b : grids_) {
			qb.push_back({{b.lo[0], b.lo[1], b.lo[2]}, {b.hi[0], b.hi[1], b.hi[2]}});
		}
		for (int d = 0; d < 3; ++d) {
			qg.domain.lo[d] = g.domain.lo[d];
			qg.domain.hi[d] = g.domain.hi[d];
			qg.periodic[d] = g.periodic[d];
		}
		qg.ndim = AMREX_SPACEDIM;
		(void)qg;
		(void)qb;
		qkhost::check(qk_ghost_plan_create(qkhost::Runtime::get().lev, &flagPlan_, &this->qgeom_, 1, 1, static_cast<int>(this->allBoxes_.size()),
						   this->allBoxes_.data(), this->owner_.data(), qkhost::Comm::get().rank),
			      "qk_ghost_plan_create(redoFlag)");
		flagPeers_.build(flagPlan_, sizeof(int));
	}

	qkhost::PeerBuffers flagPeers_;
	auto readCount() -> int64_t // cells flagged on ALL ranks: every rank takes the same branch of the FOFC / retry logic
	{
		int64_t c = 0;
		QK_HOST_HIP(hipMemcpy(&c, d_count_, sizeof(int64_t), hipMemcpyDeviceToHost));
		return qkhost::Comm::get().allReduceSum(c);
	}

	// computeHydroFluxes / hydroFluxFunction (reference src/QuokkaSimulation.hpp:1403-1517)
	template <FluxDir DIR> void hydroFluxFunction(int d)
	{
		if (reconstructionOrder_ == 3) {
			HyperbolicSystem<problem_t>::template ReconstructStatesPPM<DIR>(primVar_, leftState_[d], rightState_[d], 1, ncompHydro_);
		} else if (reconstructionOrder_ == 2) {
			HyperbolicSystem<problem_t>::template ReconstructStatesPLM<DIR, SlopeLimiter::minmod>(primVar_, leftState_[d], rightState_[d], 1, ncompHydro_);
		} else {
			HyperbolicSystem<problem_t>::template ReconstructStatesConstant<DIR>(primVar_, leftState_[d], rightState_[d], 1, ncompHydro_);
		}
		HydroSystem<problem_t>::template FlattenShocks<DIR>(primVar_, flatCoefs_[0], flatCoefs_[1], flatCoefs_[2], leftState_[d], rightState_[d], 1, ncompHydro_);
		HydroSystem<problem_t>::template ComputeFluxes<RiemannSolver::HLLC, DIR>(flux_[d], vel_[d], leftState_[d], rightState_[d], primVar_,
											 artificialViscosityK_);
	}
	void computeHydroFluxes(amrex::MultiFab const &consVar)
	{
		HydroSystem<problem_t>::ConservedToPrimitive(consVar, primVar_, nghost_cc_);
		AMREX_D_TERM(HydroSystem<problem_t>::template ComputeFlatteningCoefficients<FluxDir::X1>(primVar_, flatCoefs_[0], 2);
			     , HydroSystem<problem_t>::template ComputeFlatteningCoefficients<FluxDir::X2>(primVar_, flatCoefs_[1], 2);
			     , HydroSystem<problem_t>::template ComputeFlatteningCoefficients<FluxDir::X3>(primVar_, flatCoefs_[2], 2);)
		AMREX_D_TERM(hydroFluxFunction<FluxDir::X1>(0);, hydroFluxFunction<FluxDir::X2>(1);, hydroFluxFunction<FluxDir::X3>(2);)
	}
	template <FluxDir DIR> void hydroFOFluxFunction(int d)
	{
		HyperbolicSystem<problem_t>::template ReconstructStatesConstant<DIR>(primVar_, leftState_[d], rightState_[d], 1, ncompHydro_);
		HydroSystem<problem_t>::template ComputeFluxes<RiemannSolver::LLF, DIR>(FOflux_[d], FOvel_[d], leftState_[d], rightState_[d], primVar_,
											artificialViscosityK_);
	}
	void computeFOHydroFluxes(amrex::MultiFab const &consVar) // reference src/QuokkaSimulation.hpp:1519-1568
	{
		HydroSystem<problem_t>::ConservedToPrimitive(consVar, primVar_, nghost_cc_);
		AMREX_D_TERM(hydroFOFluxFunction<FluxDir::X1>(0);, hydroFOFluxFunction<FluxDir::X2>(1);, hydroFOFluxFunction<FluxDir::X3>(2);)
	}

	auto rhsPdvPredict(std::array<amrex::MultiFab, AMREX_SPACEDIM> const &fl, std::array<amrex::MultiFab, AMREX_SPACEDIM> const &vl,
			   amrex::MultiFab const &stateOld, amrex::MultiFab &stateNew, double dt) -> int64_t
	{
		QK_HOST_HIP(hipMemsetAsync(d_count_, 0, sizeof(int64_t), nullptr));
		HydroSystem<problem_t>::ComputeRhsFromFluxes(rhs_, fl, geom[0].CellSizeArray(), ncompHydro_);
		HydroSystem<problem_t>::AddInternalEnergyPdV(rhs_, stateOld, geom[0].CellSizeArray(), vl, redoFlag_);
		HydroSystem<problem_t>::PredictStep(stateOld, stateNew, rhs_, dt, ncompHydro_, redoFlag_, d_count_);
		return readCount();
	}

	// redoFlag.FillBoundary(geom.periodicity()) (reference src/QuokkaSimulation.hpp:1157), across ranks as the state's
	void fillFlagGhosts()
	{
		for (size_t k = 0; k < flagPeers_.peer.size(); ++k) {
			qkhost::check(qk_FillBoundary_pack_int(flagPlan_, nullptr, static_cast<int>(k), qkhost::itab(redoFlag_), static_cast<int *>(flagPeers_.send[k])),
				      "redoFlag pack");
		}
		qkhost::Comm::get().exchangeBegin(flagPeers_.peer, flagPeers_.send, flagPeers_.nsend, flagPeers_.recv, flagPeers_.nrecv, sizeof(int), nullptr);
		qkhost::check(qk_FillBoundary_local_int(flagPlan_, nullptr, qkhost::itab(redoFlag_)), "redoFlag.FillBoundary");
		qkhost::Comm::get().exchangeEnd(nullptr);
		for (size_t k = 0; k < flagPeers_.peer.size(); ++k) {
			qkhost::check(qk_FillBoundary_unpack_int(flagPlan_, nullptr, static_cast<int>(k), qkhost::itab(redoFlag_), static_cast<const int *>(flagPeers_.recv[k])),
				      "redoFlag unpack");
		}
	}

	// one RK stage exactly as the reference (src/QuokkaSimulation.hpp:1099-1198 / 1202-1287), reference-shaped operators
	bool unfusedRan_ = false; // the reference-shaped operators ran in the current advance: SyncDualEnergy leaves its flag in d_error_
	auto stageUnfused(int stageNo, amrex::MultiFab const &U_in, amrex::MultiFab const &U_old, amrex::MultiFab &U_out, double dt) -> bool
	{
		unfusedRan_ = true;
		auto *lev = qkhost::Runtime::get().lev;
		computeHydroFluxes(U_in);
		auto *fl = &flux_;
		auto *vl = &vel_;
		if (stageNo == 1) {
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				amrex::MultiFab::Copy(halfFlux_[d], flux_[d]);
				amrex::MultiFab::Copy(halfVel_[d], vel_[d]);
			}
		} else {
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				rk2flux_[d].setVal(0);
				rk2vel_[d].setVal(0);
				qkhost::check(qk_Saxpy(lev, nullptr, d, qkhost::tab(rk2flux_[d]), 0.5, qkhost::tab(halfFlux_[d]), ncompHydro_), "Saxpy");
				qkhost::check(qk_Saxpy(lev, nullptr, d, qkhost::tab(rk2vel_[d]), 0.5, qkhost::tab(halfVel_[d]), 1), "Saxpy");
				qkhost::check(qk_Saxpy(lev, nullptr, d, qkhost::tab(rk2flux_[d]), 0.5, qkhost::tab(flux_[d]), ncompHydro_), "Saxpy");
				qkhost::check(qk_Saxpy(lev, nullptr, d, qkhost::tab(rk2vel_[d]), 0.5, qkhost::tab(vel_[d]), 1), "Saxpy");
			}
			fl = &rk2flux_;
			vl = &rk2vel_;
		}
		redoFlag_.setVal(0);
		int64_t nbad = rhsPdvPredict(*fl, *vl, U_old, U_out, dt);
		if (nbad > 0) { // first-order flux correction
			++fofcStages_;
			computeFOHydroFluxes(U_old);
			fillFlagGhosts();
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				qkhost::check(qk_replaceFluxes(lev, nullptr, d, qkhost::tab((*fl)[d]), qkhost::tab(FOflux_[d]), qkhost::itab(redoFlag_), ncompHydro_),
					      "replaceFluxes");
				qkhost::check(qk_replaceFluxes(lev, nullptr, d, qkhost::tab((*vl)[d]), qkhost::tab(FOvel_[d]), qkhost::itab(redoFlag_), 1), "replaceFluxes");
			}
			nbad = rhsPdvPredict(*fl, *vl, U_old, U_out, dt);
			if (stageNo == 1 && integratorOrder_ == 1) {
				// forward Euler: halfFlux() hands halfFlux_ to incrementFluxRegisters — it must be the flux the state was updated
				// with, i.e. the CORRECTED one (the reference increments with the corrected fluxArrays, QuokkaSimulation.hpp:1160-1196)
				for (int d = 0; d < AMREX_SPACEDIM; ++d) {
					amrex::MultiFab::Copy(halfFlux_[d], flux_[d]);
				}
			}
			if (nbad > 0 && abortOnFofcFailure_ != 0) {
				return false;
			}
		}
		HydroSystem<problem_t>::EnforceLimits(densityFloor_, tempFloor_, U_out);
		if (useDualEnergy_ == 1) {
			HydroSystem<problem_t>::SyncDualEnergy(U_out, d_error_);
		}
		// (stage 2: *fl IS rk2flux_, possibly FOFC-corrected — where the fused stage leaves flux_rk2 as well)
		return true;
	}

	// ------------------------------------------------------------------ the fused stage (qk_hydro_stage_fused) and the multi-GPU schedule
	// the fused stage carries up to 3 passive scalars; mass scalars take the reference-shaped operators
	[[nodiscard]] static constexpr auto fusedEligible() -> bool
	{
		// (any AMREX_SPACEDIM since round 3: in a 1-D build the x sweep carries the epilogue, in a 2-D build the y sweep)
		return HydroSystem<problem_t>::nscalars_ <= 3 && Physics_Traits<problem_t>::numMassScalars == 0;
	}
	[[nodiscard]] auto isFinalStage(int stageNo) const -> bool { return (stageNo == 2) || (integratorOrder_ == 1); }
	// the carried-right-hand-side form of the RK2 average (qk_hydro_stage_args::rk2_carry_rhs; deck: hydro.rk2_carry_rhs = 1, default 0): only
	// where nothing consumes flux_rk2 (no flux registers) and the integrator has two stages
	[[nodiscard]] auto carryActive() const -> bool { return AMREX_SPACEDIM == 3 && rk2CarryRhs_ != 0 && integratorOrder_ == 2 && !storeFluxRk2_ && !forceExactForm_; }
	bool forceExactForm_ = false; // (set while stage 2 of the carried form is redone in the exact form: correctStage)

	// Boxes of this rank in two groups for the overlapped ghost fill: [0] early — every ghost cell is filled on this GPU —, [1] late — waits
	// for strips from other ranks (qk_ghost_plan_box_is_remote).  Each group is a sub-level whose descriptor tables alias the level's arrays.
	struct OverlapGroup {
		std::vector<int> idx;
		qk_level *lev = nullptr;
		std::map<const void *, void *> tables; // descriptor table of a MultiFab -> the same descriptors of this group's boxes, contiguous
	};
	OverlapGroup groups_[2];
	// the speculative coarse step (advanceLevelBegin): near / far sub-levels, the far boxes' launch sets, their stream, scratch and words
	OverlapGroup specNear_, specFar_;
	std::vector<OverlapGroup> specFarParts_;
	hipStream_t farStream_ = nullptr;
	hipEvent_t evInterReady_ = nullptr, evFarDone_ = nullptr;
	void *farScratch_ = nullptr;
	int64_t farScratchBytes_ = 0;
	double *d_fixFar_ = nullptr;
	bool fixFarPending_ = false;
	static void releaseGroup(OverlapGroup &G)
	{
		for (auto &kv : G.tables) {
			(void)hipFree(kv.second);
		}
		G.tables.clear();
		if (G.lev != nullptr) {
			qk_level_destroy(G.lev);
			G.lev = nullptr;
		}
		G.idx.clear();
	}
	int overlapState_ = 0; // 0: not examined, 1: active, 2: not worth it / not possible
	amrex::Long minOverlapCells_ = 8L * 128 * 128 * 128; // a launch fills the chip from ~8 boxes of 128^3 on (the marching sweeps expose one wave per 64 cells of a pencil)

	auto overlapActive() -> bool
	{
		if (overlapState_ == 0) {
			overlapState_ = 2;
			amrex::ParmParse pq("qk");
			pq.query("min_overlap_cells", minOverlapCells_); // (tests: 1 forces the split on small problems)
			if (!this->beforePhysBC_ && fusedEligible()) {
				amrex::Long cells[2] = {0, 0};
				for (int b = 0; b < static_cast<int>(grids_.size()); ++b) {
					int const g = (qk_ghost_plan_box_is_remote(this->plan_, b) == 1) ? 1 : 0;
					groups_[g].idx.push_back(b);
					cells[g] += grids_[b].numPts();
				}
				if (!groups_[0].idx.empty() && !groups_[1].idx.empty() && std::min(cells[0], cells[1]) >= minOverlapCells_) {
					for (auto &g : groups_) {
						std::vector<qk_box> qb;
						for (int b : g.idx) {
							qb.push_back({{grids_[b].lo[0], grids_[b].lo[1], grids_[b].lo[2]}, {grids_[b].hi[0], grids_[b].hi[1], grids_[b].hi[2]}});
						}
						qkhost::check(qk_level_create(qkhost::Runtime::get().ctx, &g.lev, AMREX_SPACEDIM, static_cast<int>(qb.size()), qb.data()), "qk_level_create");
					}
					overlapState_ = 1;
					std::cout << "rank " << qkhost::Comm::get().rank << ": overlapped ghost fill: " << groups_[0].idx.size() << " early / " << groups_[1].idx.size()
						  << " late boxes\n";
				}
			}
		}
		return overlapState_ == 1;
	}
	// the descriptors of group g's boxes out of a MultiFab's table (64 bytes each), gathered once per table
	template <typename D> auto groupTable(int g, D *full) -> D * { return groupTableOf(groups_[g], full); }
	template <typename D> static auto groupTableOf(OverlapGroup &G, D *full) -> D *
	{
		if (full == nullptr) {
			return nullptr;
		}
		auto &m = G.tables;
		auto it = m.find(full);
		if (it == m.end()) {
			void *p = nullptr;
			QK_HOST_HIP(hipMalloc(&p, sizeof(D) * G.idx.size()));
			for (size_t n = 0; n < G.idx.size(); ++n) {
				QK_HOST_HIP(hipMemcpy(static_cast<D *>(p) + n, full + G.idx[n], sizeof(D), hipMemcpyDeviceToDevice));
			}
			it = m.emplace(full, p).first;
		}
		return static_cast<D *>(it->second);
	}

	void fusedBegin(int stageNo, bool bothSlots = false)
	{
		hipStream_t const cs = qkhost::Runtime::get().computeStream();
		(void)stageNo;
		// (sig0, sig1, count of slot 0 — or of both slots; the error flags are sticky: a set flag ends the run)
		QK_HOST_HIP(hipMemsetAsync(d_words_, 0, 3 * sizeof(int64_t), cs));
		if (bothSlots) {
			QK_HOST_HIP(hipMemsetAsync(d_words_ + 4, 0, 3 * sizeof(int64_t), cs));
		}
	}
	// both slots in ONE blocking device -> host copy; with several ranks ONE all-reduce(MAX) (a count is only ever compared with zero)
	void readWords(StageWords w[2])
	{
		int64_t h[8];
		QK_HOST_HIP(hipMemcpy(h, d_words_, sizeof(h), hipMemcpyDeviceToHost));
		double v[8];
		for (int sl = 0; sl < 2; ++sl) {
			std::memcpy(&v[4 * sl], &h[4 * sl], 2 * sizeof(double));
			v[4 * sl + 2] = static_cast<double>(h[4 * sl + 2]);
			v[4 * sl + 3] = static_cast<double>(h[4 * sl + 3] & 0xFFFFFFFFLL);
		}
		if (qkhost::Comm::get().size > 1) {
			qkhost::Comm::get().allReduce(v, 8, qkhost::Comm::Op::max);
		}
		for (int sl = 0; sl < 2; ++sl) {
			w[sl].sig[0] = v[4 * sl];
			w[sl].sig[1] = v[4 * sl + 1];
			w[sl].count = static_cast<int64_t>(v[4 * sl + 2]);
			w[sl].err = static_cast<int64_t>(v[4 * sl + 3]);
		}
	}
	// one fused stage over all local boxes (group < 0) or over one group of the overlapped fill
	// grp / scratch / stream: a sub-level of its own with its own stage scratch on its own stream (the far boxes of a speculative coarse step)
	void fusedLaunch(int stageNo, amrex::MultiFab const &U_in, amrex::MultiFab const &U_old, amrex::MultiFab &U_out, double dt, int group = -1, bool fofc = false,
			 int slot = 0, OverlapGroup *grp = nullptr, void *scratch = nullptr, int64_t scratchBytes = 0, hipStream_t stream = nullptr)
	{
		auto t = qkhost::traits<problem_t>();
		if (grp == nullptr && group >= 0) {
			grp = &groups_[group];
		}
		auto sel = [&](qk_array4 *full) { return grp == nullptr ? full : groupTableOf(*grp, full); };
		qk_hydro_stage_args a{};
		a.U_in = sel(qkhost::tab(U_in));
		a.U_old = sel(qkhost::tab(U_old));
		a.U_out = sel(qkhost::tab(U_out));
		for (int d = 0; d < 3; ++d) {
			a.halfFlux[d] = (d < AMREX_SPACEDIM) ? sel(qkhost::tab(halfFlux_[d])) : nullptr;
			a.halfVel[d] = (d < AMREX_SPACEDIM) ? sel(qkhost::tab(halfVel_[d])) : nullptr;
			a.dx[d] = (d < AMREX_SPACEDIM) ? geom[0].dx[d] : 1.0;
		}
		a.redoFlag = grp == nullptr ? qkhost::itab(redoFlag_) : groupTableOf(*grp, qkhost::itab(redoFlag_));
		a.d_redo_count = d_words_ + 4 * slot + 2;
		a.d_error_flag = reinterpret_cast<int *>(d_words_ + 4 * slot + 3);
		if (isFinalStage(stageNo)) {
			a.d_max_signal = reinterpret_cast<double *>(d_words_ + 4 * slot);
		}
		a.scratch = (scratch != nullptr) ? scratch : scratch_;
		a.scratch_bytes = (scratch != nullptr) ? scratchBytes : scratchBytes_;
		a.dt = dt;
		a.stage = stageNo;
		a.reconstruction_order = reconstructionOrder_;
		a.densityFloor = densityFloor_;
		a.tempFloor = tempFloor_;
		a.use_dual_energy = useDualEnergy_;
		a.K_visc = artificialViscosityK_;
		bool const masked = carryActive() && fluxMask_.size() > 0;
		a.store_flux_rk2 = (!carryActive() && needsFluxRk2()) ? 1 : 0; // (also the carried form forced into the exact one for a stage-2 correction)
		for (int d = 0; d < 3; ++d) {
			a.fluxRk2[d] = ((a.store_flux_rk2 != 0 || masked) && d < AMREX_SPACEDIM) ? sel(qkhost::tab(rk2flux_[d])) : nullptr;
		}
		if (masked) {
			auto *full = reinterpret_cast<qk_carray4 *>(fluxMask_.arrays());
			a.flux_mask = grp == nullptr ? full : groupTableOf(*grp, full);
		}
		if (carryActive()) {
			if (rhs1_.size() == 0) {
				rhs1_.define(grids_, ncompHydro_ + 1, 0);
			}
			a.rk2_carry_rhs = 1;
			a.rhs1 = sel(qkhost::tab(rhs1_));
		}
		a.fofc_pass = fofc ? 1 : 0;
		if (primNow_ && !fofc) { // the primitive hand-off between the two stages of this step (stagePairSpeculative)
			a.prim_out = (stageNo == 1) ? 1 : 0;
			a.prim_in = (stageNo == 2) ? 1 : 0;
		}
		qkhost::check(qk_hydro_stage_fused(grp == nullptr ? qkhost::Runtime::get().lev : grp->lev, stream != nullptr ? stream : qkhost::Runtime::get().computeStream(), &t, &a),
			      "qk_hydro_stage_fused");
	}
	// flagged cells of the stage (all ranks); after a clean final stage the two CFL maxima are kept for computeTimestep / isCflViolated
	auto fusedEnd(int stageNo, StageWords const *given = nullptr) -> int64_t
	{
		StageWords w[2];
		if (given == nullptr) {
			readWords(w);
			given = &w[0];
		}
		if (given->err != 0) {
			amrex::Abort("density is negative in SyncDualEnergy! abort!!");
		}
		int64_t const nbad = given->count;
		if (nbad == 0) {
			stage1LeftF1_ = (stageNo == 1) ? !carryActive() : stage1LeftF1_;
			if (isFinalStage(stageNo)) {
				signal_[0] = given->sig[0];
				signal_[1] = given->sig[1];
				haveSignal_ = true;
			}
		}
		return nbad;
	}
	// a stage whose fused attempt flagged cells, on the reference-shaped operators.  Stage 2 forms 0.5 F1 + 0.5 F2 from halfFlux_: after a fused
	// stage 1 in the carried-rhs mode F1 was never stored and is evaluated again from the old state (ghost cells still filled; same device
	// functions, same values)
	auto redoStageUnfused(int stageNo, amrex::MultiFab const &U_in, amrex::MultiFab const &U_old, amrex::MultiFab &U_out, double dt) -> bool
	{
		if (stageNo == 2 && !stage1LeftF1_) {
			computeHydroFluxes(U_old);
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				amrex::MultiFab::Copy(halfFlux_[d], flux_[d]);
				amrex::MultiFab::Copy(halfVel_[d], vel_[d]);
			}
		}
		if (stageNo == 1) {
			stage1LeftF1_ = true;
		}
		return stageUnfused(stageNo, U_in, U_old, U_out, dt);
	}

	// the fused first pass of a stage flagged cells: first-order flux correction (reference src/QuokkaSimulation.hpp:1144-1184, :1232-1270) as ONE
	// more fused pass (qk_hydro_stage_args::fofc_pass) where it applies — no artificial viscosity, not stage 2 of the carried-rhs form, not forward
	// Euler feeding flux registers (whose halfFlux must hold the corrected flux) —, the whole stage on the operators otherwise
	auto correctStage(int stageNo, amrex::MultiFab const &U_in, amrex::MultiFab const &U_old, amrex::MultiFab &U_out, double dt) -> bool
	{
		if (stageNo == 1) {
			stage1LeftF1_ = !carryActive();
		}
		bool const applies = fusedFofc_ != 0 && artificialViscosityK_ == 0.0 && !(integratorOrder_ == 1 && storeFluxRk2_);
		if (!applies) {
			return redoStageUnfused(stageNo, U_in, U_old, U_out, dt);
		}
		if (carryActive() && stageNo == 2) {
			// The correction replaces flux_rk2 = 0.5 F1 + 0.5 F2 of a face as a whole, and the carried form never stored F1: the stage is redone in
			// the reference's form, still on the fused kernels — the stage-1 sweeps once more over the old state (ghost cells still filled) to
			// leave F1 in halfFlux_ (the state they write is discarded), stage 2 in the exact form, then its correction pass.
			++fofcStages_;
			forceExactForm_ = true;
			fusedBegin(1);
			fusedLaunch(1, U_old, U_old, U_out, dt);
			stage1LeftF1_ = true;
			fusedBegin(2);
			fusedLaunch(2, U_in, U_old, U_out, dt);
			int64_t nbad = fusedEnd(2);
			if (nbad > 0) {
				fillFlagGhosts();
				fusedBegin(2);
				fusedLaunch(2, U_in, U_old, U_out, dt, -1, true);
				nbad = fusedEnd(2);
			}
			forceExactForm_ = false;
			return !(nbad > 0 && abortOnFofcFailure_ != 0);
		}
		++fofcStages_;
		fillFlagGhosts();
		fusedBegin(stageNo);
		fusedLaunch(stageNo, U_in, U_old, U_out, dt, -1, true);
		return !(fusedEnd(stageNo) > 0 && abortOnFofcFailure_ != 0);
	}

	auto stage(int stageNo, amrex::MultiFab const &U_in, amrex::MultiFab const &U_old, amrex::MultiFab &U_out, double dt) -> bool
	{
		if constexpr (fusedEligible()) {
			fusedBegin(stageNo);
			fusedLaunch(stageNo, U_in, U_old, U_out, dt);
			if (fusedEnd(stageNo) == 0) {
				return true;
			}
			return correctStage(stageNo, U_in, U_old, U_out, dt);
		}
		return redoStageUnfused(stageNo, U_in, U_old, U_out, dt);
	}

	// fillBoundaryConditions(U_in) + one RK stage.  With boxes on other ranks the early group is advanced while the strips of the late group are
	// on the wire (north_star: FillBoundary overlapped with the update on a second HIP stream — RCCL's); the reference's fill is blocking
	// (src/QuokkaSimulation.hpp:1076, :1204).  Same arithmetic per cell: bit-identical to the blocking schedule.
	auto fillAndStage(int stageNo, amrex::MultiFab &U_in, amrex::MultiFab const &U_old, amrex::MultiFab &U_out, double dt) -> bool
	{
		if constexpr (fusedEligible()) {
			if (overlapActive()) {
				fusedBegin(stageNo);
				launchStage(stageNo, U_in, U_old, U_out, dt, 0);
				if (fusedEnd(stageNo) == 0) {
					return true;
				}
				return correctStage(stageNo, U_in, U_old, U_out, dt);
			}
		}
		this->fillBoundaryConditions(U_in);
		return stage(stageNo, U_in, U_old, U_out, dt);
	}
	// fillBoundaryConditions(U_in) + the fused launches of one stage (early / late groups where boxes wait for other ranks), nothing read back
	void launchStage(int stageNo, amrex::MultiFab &U_in, amrex::MultiFab const &U_old, amrex::MultiFab &U_out, double dt, int slot)
	{
		if (overlapActive()) {
			this->fillBoundaryConditions(U_in, [&]() { fusedLaunch(stageNo, U_in, U_old, U_out, dt, 0, false, slot); });
			fusedLaunch(stageNo, U_in, U_old, U_out, dt, 1, false, slot);
			return;
		}
		this->fillBoundaryConditions(U_in);
		fusedLaunch(stageNo, U_in, U_old, U_out, dt, -1, false, slot);
	}
	// Both stages of an RK2 step enqueued before either redo count is read: the GPU does not idle through device -> host round trips between the
	// stages.  Stage 2 is speculative — if stage 1 flagged cells (rare) its work is discarded and the caller redoes the stages in order; the old
	// state is untouched by either stage, so the result is the same.  Returns 1: both stages done, 0: the step failed, -1: redo in order.
	auto stagePairSpeculative(amrex::MultiFab &U_old, double time, double dt) -> int
	{
		if (primBackoff_ > 0) { // (a recent attempt was dropped: see below)
			--primBackoff_;
		} else if (primHandoffApplies()) {
			// The primitive hand-off (qk_hydro_stage_args::prim_out / prim_in): stage 1 stores the primitives of the intermediate state, stage 2
			// reads them — no conversion in its pre-pass and sweeps, same bytes, same bits.  It has no correction pass: when either stage flags
			// a cell the attempt is dropped (the old state is untouched by both stages) and the step proceeds below as it does without it.
			primNow_ = true;
			fusedBegin(1, true);
			fillTime_ = time;
			launchStage(1, U_old, U_old, state_inter_cc_, dt, 0);
			fillTime_ = time + dt;
			launchStage(2, state_inter_cc_, U_old, state_new_cc_[0], dt, 1);
			primNow_ = false;
			StageWords w[2];
			readWords(w);
			if (w[0].count == 0 && w[1].count == 0) {
				fusedEnd(1, &w[0]);
				fusedEnd(2, &w[1]);
				primBackoffLen_ = 0;
				return 1;
			}
			++primHandoffDropped_;
			// a flow that flags cells step after step would pay both stages twice every time: the hand-off sits out the next 4, 8, ... 64
			// advances after a drop and comes back after a clean attempt
			primBackoffLen_ = std::min(64, std::max(4, 2 * primBackoffLen_));
			primBackoff_ = primBackoffLen_;
			invalidateSignal();
		}
		fusedBegin(1, true);
		fillTime_ = time;
		launchStage(1, U_old, U_old, state_inter_cc_, dt, 0);
		fillTime_ = time + dt;
		launchStage(2, state_inter_cc_, U_old, state_new_cc_[0], dt, 1);
		StageWords w[2];
		readWords(w);
		if (fusedEnd(1, &w[0]) != 0) {
			return -1;
		}
		if (fusedEnd(2, &w[1]) == 0) {
			return 1;
		}
		return correctStage(2, state_inter_cc_, U_old, state_new_cc_[0], dt) ? 1 : 0;
	}
	// The primitive hand-off applies to a hydro level without coarse-fine ghost cells (a plain level, or level 0 of a hierarchy: its children
	// interpolate from its old and new states, never from the intermediate one): no radiation variables in the state, gamma law with
	// reconstruct_eint off, and boundary rules that act component by component (reflect / extrapolate / periodic: the momenta's parity is the
	// velocities'); a problem with ext_dir faces writes CONSERVED values through its functor.
	[[nodiscard]] auto primHandoffApplies() -> bool
	{
		if (primHandoffChecked_) {
			return primHandoffOk_;
		}
		// customBcKernel runs a specialised setCustomBoundaryConditions on every ghost slab whatever the BCRec says: a problem that specialises
		// the hook writes CONSERVED values — the hand-off waits until the first fill has shown that the hook is the empty default
		if (this->customBcIsDefault_ < 0) {
			return false; // (not known yet: ask again after the first fill)
		}
		primHandoffChecked_ = true;
		int on = 1;
		amrex::ParmParse("qk").query("prim_handoff", on);
		auto const t = qkhost::traits<problem_t>();
		bool ok = on != 0 && this->customBcIsDefault_ == 1 && this->amrLevel_ == 0 && !is_radiation_enabled_ && t.reconstruct_eint == 0 &&
			  t.eos_temperature_model == 0 && !(t.cs_isothermal == t.cs_isothermal) && t.gamma != 1.0 &&
			  Physics_Indices<problem_t>::nvarTotal_cc == ncompHydro_;
		for (auto const &bc : this->BCs_cc_) {
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				ok = ok && bc.lo(d) != amrex::BCType::ext_dir && bc.hi(d) != amrex::BCType::ext_dir && bc.lo(d) != amrex::BCType::hoextrap &&
				     bc.hi(d) != amrex::BCType::hoextrap;
			}
		}
		primHandoffOk_ = ok;
		return ok;
	}
	bool primHandoffChecked_ = false, primHandoffOk_ = false, primNow_ = false;
	long primHandoffDropped_ = 0;
	int primBackoff_ = 0, primBackoffLen_ = 0;
	int rk2CarryRhs_ = 0;	   // deck: hydro.rk2_carry_rhs
	int fusedFofc_ = 1;	   // deck: qk.fused_fofc (0: a flagged stage is redone on the reference-shaped operators; tests)
	bool stage1LeftF1_ = true; // halfFlux_ holds the stage-1 fluxes of the current step
	amrex::MultiFab rhs1_;
};

template <typename problem_t> void QuokkaSimulation<problem_t>::preCalculateInitialConditions() {}
template <typename problem_t> void QuokkaSimulation<problem_t>::setInitialConditionsOnGridFaceVars(quokka::grid const & /*grid_elem*/) {}

// (a problem that specialises one of these hooks never sets the flag: the driver then drops its cached signal speeds after the call)
template <typename problem_t> void QuokkaSimulation<problem_t>::computeAfterTimestep() { afterTimestepIsDefault_ = true; }
template <typename problem_t> void QuokkaSimulation<problem_t>::computeBeforeTimestep() { beforeTimestepIsDefault_ = true; }
template <typename problem_t> void QuokkaSimulation<problem_t>::createInitialParticles() {}
template <typename problem_t>
void QuokkaSimulation<problem_t>::computeAfterLevelAdvance(int /*lev*/, amrex::Real /*time*/, amrex::Real /*dt_lev*/, int /*ncycle*/)
{
	afterLevelAdvanceIsDefault_ = true;
}
template <typename problem_t>
void QuokkaSimulation<problem_t>::ComputeDerivedVar(int /*lev*/, std::string const & /*dname*/, amrex::MultiFab & /*mf*/, const int /*ncomp*/) const
{
}
template <typename problem_t>
auto QuokkaSimulation<problem_t>::ComputeProjections(int /*dir*/) const -> std::unordered_map<std::string, amrex::BaseFab<amrex::Real>>
{
	return std::unordered_map<std::string, amrex::BaseFab<amrex::Real>>{};
}
template <typename problem_t> auto QuokkaSimulation<problem_t>::ComputeStatistics() -> std::map<std::string, amrex::Real> { return std::map<std::string, amrex::Real>{}; }
template <typename problem_t>
void QuokkaSimulation<problem_t>::addStrangSplitSources(amrex::MultiFab & /*state*/, int /*lev*/, amrex::Real /*time*/, amrex::Real /*dt_lev*/)
{
	strangSourcesAreDefault_ = true; // (a problem that specialises the hook never sets this)
}

// generic computeAfterEvolve: relative rms L1 error norm vs the problem's reference solution (reference src/QuokkaSimulation.hpp:620-644)
template <typename problem_t> void QuokkaSimulation<problem_t>::computeAfterEvolve(amrex::Vector<amrex::Real> & /*initSumCons*/)
{
	if (!computeReferenceSolution_) {
		return;
	}
	int const ncomp = state_new_cc_[0].nComp();
	amrex::MultiFab ref(grids_, ncomp, 0);
	computeReferenceSolution(ref, geom[0].CellSizeArray(), geom[0].ProbLoArray());
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
		sol_norm += rn * rn;
		err_norm += en * en;
	}
	errorNorm_ = std::sqrt(err_norm) / std::sqrt(sol_norm);
	amrex::Print() << "Relative rms L1 error norm = " << errorNorm_ << "\n";
}

// test hook: `qk.dump_state = <file>` writes the valid cells of state_new_cc_ (double, [box][comp][k][j][i]) after evolve()
inline void qkDumpFields(amrex::MultiFab &mf, int istep, double tNew, double dt, long fofcStages, long retries, double errorNorm)
{
	std::string path;
	amrex::ParmParse pp("qk");
	if (!pp.query("dump_state", path)) {
		return;
	}
	if (qkhost::Comm::get().size > 1) { // one file per rank: its boxes in global order
		path += ".rank" + std::to_string(qkhost::Comm::get().rank);
	}
	std::ofstream f(path, std::ios::binary);
	for (int b = 0; b < mf.size(); ++b) {
		auto h = mf.copyToHost(b);
		amrex::Array4<double> a(h.data(), mf.fabbox(b), mf.nComp());
		auto const &vb = mf.validbox(b);
		for (int n = 0; n < mf.nComp(); ++n) {
			for (int k = vb.lo[2]; k <= vb.hi[2]; ++k) {
				for (int j = vb.lo[1]; j <= vb.hi[1]; ++j) {
					f.write(reinterpret_cast<const char *>(&a(vb.lo[0], j, k, n)), static_cast<std::streamsize>(sizeof(double)) * vb.length(0));
				}
			}
		}
	}
	std::ofstream meta(path + ".meta");
	meta.precision(17);
	meta << istep << " " << tNew << " " << dt << " " << fofcStages << " " << retries << " " << errorNorm << "\n";
}
template <typename problem_t> void qkDumpState(QuokkaSimulation<problem_t> &sim)
{
	qkDumpFields(sim.state_new_cc_[0], sim.istep[0], sim.tNew_[0], sim.dt_[0], sim.fofcStages_, sim.retries_, sim.errorNorm_);
}

// every problem executable: amrex::Initialize analogue + problem_main()
auto problem_main() -> int;
#ifndef QK_HOST_NO_MAIN
int main(int argc, char **argv)
{
	amrex::ParmParse::Initialize(argc, argv);
	int const rc = problem_main();
	return rc;
}
#endif

#endif // QK_HOST_QUOKKA_HOST_HPP_
