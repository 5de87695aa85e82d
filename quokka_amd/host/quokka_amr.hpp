// quokka_amr.hpp — the AMR level machinery of AMRSimulation / QuokkaSimulation for the C++ host mirror:
//   timeStepWithSubcycling, regrid, computeTimestep over levels   reference src/simulation.hpp:744-818,1220-1343
//   FillPatch on refined levels                                     src/simulation.hpp:1704-1858
//   incrementFluxRegisters / Reflux / AverageDownTo / FixupState    :1345-1387, :1308, :1949-1964, src/QuokkaSimulation.hpp:761-770
// Level l of the hierarchy is one QuokkaSimulation<problem_t> object built from a LevelSpec (its own geometry, boxes, state, ghost
// plan, stage scratch); the base object (level 0, built from the deck) owns the finer ones through this driver.  Every cell is
// touched by a kernel behind include/quokka_amd.h; the grids come from qk_amr_tile_flags + qk_amr_cluster_tiles (tile clustering,
// not AMReX's Berger-Rigoutsos: unpinned).  Same algorithm as quokka_amd/amr_simulation.py, which the GPU tests pin by
// properties (full-coverage == uniform fine run, conservation with reflux, nesting).
// Several ranks (one process per GPU): the level-0 boxes are distributed by the base object; a refined box lives on the rank of its level-0
// ancestor, so grids are clustered inside each level-0 box, and interpolation, average-down and regrid copies stay on the GPU that owns the
// data.  Only the ordinary ghost exchange of every level and the reflux increments (register cells next to a box of another rank: folded with
// qk_SumBoundary_*) cross ranks; tile flags are all-reduced so that every rank builds the same hierarchy.
// qk.distribute_levels = 1: every level has its own box -> rank map instead, as AMReX hands AMRSimulation a BoxArray + DistributionMapping per level
// (reference src/simulation.hpp:1421-1500, :1657-1702) — grids clustered globally as with one rank, a level with fewer boxes than ranks chopped until
// every rank can own one (qk_grid_layout.hpp: chopGrids, distributeSfc), and the coarse data a refined level needs and produces kept on the FINE
// level's ranks (Shadow below: the coarse patch of FillPatchTwoLevels, the coarsened fine array of average_down, the cfpatch of YAFluxRegister), moved
// between the two distributions by ParallelCopy plans (qk_pcopy.hpp).  The schedule of quokka_amd/amr_simulation.py (CoarseShadow, DistFluxRegister).
#ifndef QK_HOST_QUOKKA_AMR_HPP_
#define QK_HOST_QUOKKA_AMR_HPP_

#include <chrono>
#include <map>
#include <memory>

#include "qk_grid_layout.hpp"
#include "qk_pcopy.hpp"
#include "quokka_host.hpp"

// SimT: the simulation class of one level — QuokkaSimulation<problem_t> (hydro / radiation hydrodynamics) or AdvectionSimulation<problem_t>
// (quokka_advection.hpp: SimT::isAdvection; no retries, no energy hooks in the interpolation, both RK stages feed the flux registers)
template <typename problem_t, typename SimT> class AmrDriver
{
      public:
	using Sim = SimT;

	explicit AmrDriver(Sim &base) : base_(base)
	{
		amrex::ParmParse pa("amr");
		pa.query("max_level", max_level);
		pa.query("blocking_factor", blocking_factor);
		pa.query("n_error_buf", n_error_buf);
		pa.query("regrid_int", regrid_int);
		std::vector<int> mgs;
		if (pa.queryarr("max_grid_size", mgs) && !mgs.empty()) {
			max_grid_size = mgs[0];
		}
		amrex::ParmParse pp;
		pp.query("do_reflux", do_reflux);
		pp.query("grid_eff", grid_eff);
		multi_ = qkhost::Comm::get().size > 1;
		amrex::ParmParse pq("qk");
		distLevels_ = multi_ ? 1 : 0; // several ranks: every level with its own box -> rank map, as AMReX distributes them (0: refined boxes stay on the rank
		pq.query("distribute_levels", distLevels_); // of their level-0 ancestor — the round-5 scheme; 1 on one rank: the same plans, same-rank items only)
		pq.query("refine_grid_layout_target", refineTarget_); // (tests: one rank building the grids N ranks build)
		clusterWithinParent_ = (multi_ && distLevels_ == 0) ? 1 : 0;
		pq.query("cluster_within_parent", clusterWithinParent_); // (tests: one rank building the grids several ranks build)
		AMREX_ALWAYS_ASSERT(!multi_ || distLevels_ != 0 || clusterWithinParent_ != 0);
		AMREX_ALWAYS_ASSERT(distLevels_ == 0 || clusterWithinParent_ == 0);
		istep.assign(max_level + 1, 0);
		last_regrid_step.assign(max_level + 1, 0);
		dt_.assign(max_level + 1, 1.e100);
		cellUpdatesEachLevel_.assign(max_level + 1, 0);
		if constexpr (!Sim::isAdvection) {
			base_.storeFluxRk2_ = (max_level > 0);
		}
	}

	int max_level = 0, blocking_factor = 8, n_error_buf = 1, regrid_int = 2, max_grid_size = 128, do_reflux = 1;
	double grid_eff = 0.7; // AMReX's default
	int amrInterpMethod_ = 1;
	std::vector<int> istep, last_regrid_step;
	std::vector<double> dt_;
	std::vector<amrex::Long> cellUpdatesEachLevel_;
	amrex::Long cellUpdates_ = 0;
	double tNew_ = 0.0, elapsedSeconds_ = 0.0;
	bool multi_ = false;
	int clusterWithinParent_ = 0;
	int distLevels_ = 0;	 // qk.distribute_levels
	int refineTarget_ = -1; // qk.refine_grid_layout_target: the box count ChopGrids aims for (default: the number of ranks)

	[[nodiscard]] auto finestLevel() const -> int { return static_cast<int>(finer_.size()); }
	auto level(int l) -> Sim & { return l == 0 ? base_ : *finer_[l - 1]->sim; }
	// amrex::average_down of an auxiliary cell-centred field living on the grids of levels crseLev + 1 and crseLev
	void averageDownField(int crseLev, amrex::MultiFab const &fine, amrex::MultiFab &crse)
	{
		Finer &f = *finer_[crseLev];
		if (f.shadow) { // averaged on the fine boxes' ranks, then to the owners of the coarse cells
			amrex::MultiFab tmp(f.shadow->mine, crse.nComp(), 0);
			qkhost::check(qk_average_down(f.avgdown, nullptr, qkhost::tab(fine), qkhost::tab(tmp), 0, crse.nComp()), "qk_average_down");
			f.shadow->toParent(crse.nComp())(tmp, crse);
			return;
		}
		qkhost::check(qk_average_down(f.avgdown, nullptr, qkhost::tab(fine), qkhost::tab(crse), 0, crse.nComp()), "qk_average_down");
	}

	// AmrCore::InitFromScratch + AverageDown (reference src/simulation.hpp:1656-1702)
	void setInitialConditions()
	{
		base_.AMRSimulation<problem_t>::setInitialConditions(); // level 0: problem ICs (or the checkpoint's level 0), ghost cells, state_old = state_new
		if (!base_.restart_chkfile.empty()) {
			readCheckpointLevels();
			return;
		}
		for (int lev = 0; lev < max_level; ++lev) {
			auto boxes = chop(lev + 1, newGrids(lev, nullptr));
			if (boxes.empty()) {
				break;
			}
			makeLevel(lev + 1, boxes);
			level(lev + 1).setInitialConditionsAtLevel();
		}
		// AmrMesh::MakeNewGrids(time) iterates at start-up: once a level exists, the levels below it are rebuilt around it (top-down, with the
		// nesting footprints), which may make room for one more level — a level whose first grids were too narrow to hold a child (the ring
		// around a sharp pulse: Advection2D) gets its child in the next pass.  Every (re)built level takes the problem's initial conditions.
		for (int pass = 0; pass <= max_level; ++pass) {
			std::vector<std::vector<amrex::Box>> before;
			for (int l = 1; l <= finestLevel(); ++l) {
				before.push_back(level(l).allGrids_);
			}
			regrid(0);
			bool same = static_cast<int>(before.size()) == finestLevel();
			for (int l = 1; l <= finestLevel(); ++l) {
				level(l).setInitialConditionsAtLevel();
				same = same && sameBoxes(before[l - 1], level(l).allGrids_);
			}
			if (same) {
				break;
			}
		}
		for (int lev = finestLevel() - 1; lev >= 0; --lev) {
			averageDownTo(lev);
		}
	}

	// QK_AMR_HOSTPROF=1: wall time of the phases of a coarse step, the device drained before and after each (so the phases do not overlap and the
	// total is larger than an unprofiled run's); printed after the figure of merit.  A debugging aid for the host side of the hierarchy.
	struct Phase {
		Phase(AmrDriver &d, std::string name) : d_(d), name_(std::move(name)), on_(d.hostProf_)
		{
			if (on_) {
				(void)hipDeviceSynchronize();
				t0_ = std::chrono::steady_clock::now();
			}
		}
		~Phase()
		{
			if (on_) {
				(void)hipDeviceSynchronize();
				d_.phaseSeconds_[name_] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
			}
		}
		Phase(Phase const &) = delete;
		auto operator=(Phase const &) -> Phase & = delete;
		AmrDriver &d_;
		std::string name_;
		bool on_;
		std::chrono::steady_clock::time_point t0_;
	};
	bool hostProf_ = std::getenv("QK_AMR_HOSTPROF") != nullptr;
	std::map<std::string, double> phaseSeconds_;

	void evolve()
	{
		int const nc = Physics_Indices<problem_t>::nvarTotal_cc;
		double const vol = AMREX_D_TERM(base_.geom[0].dx[0], *base_.geom[0].dx[1], *base_.geom[0].dx[2]);
		amrex::Vector<amrex::Real> init_sum_cons(nc);
		for (int n = 0; n < nc; ++n) { // level 0 holds the average of every finer level: its sum is the composite integral
			init_sum_cons[n] = base_.state_new_cc_[0].sum(n) * vol;
		}
		QK_HOST_HIP(hipDeviceSynchronize());
		auto const t0 = std::chrono::steady_clock::now();
		tNew_ = base_.tNew_[0];
		int const debugMaxSteps = (std::getenv("QK_MAX_COARSE_STEPS") != nullptr) ? std::atoi(std::getenv("QK_MAX_COARSE_STEPS")) : -1; // (debugging aid)
		while (istep[0] < base_.maxTimesteps_ && tNew_ < base_.stopTime_ && (debugMaxSteps < 0 || istep[0] < debugMaxSteps)) {
			{
				Phase const ph(*this, "computeTimestep");
				computeTimestep();
			}
			if constexpr (!Sim::isAdvection) {
				base_.callBeforeTimestep(); // reference src/simulation.hpp:864-867
				dropSignalsIfHooked(base_.beforeTimestepIsDefault_);
			}
			timeStepWithSubcycling(0, tNew_);
			tNew_ += dt_[0];
			base_.tNew_[0] = tNew_;
			base_.dt_[0] = dt_[0];
			base_.istep[0] = istep[0];
			if constexpr (!Sim::isAdvection) {
				base_.callAfterTimestep(); // reference src/simulation.hpp:890
				dropSignalsIfHooked(base_.afterTimestepIsDefault_);
				base_.outputAfterStep(istep[0] - 1);
			}
			if (tNew_ >= base_.stopTime_ - 1.e-6 * dt_[0]) {
				break;
			}
			if constexpr (!Sim::isAdvection) {
				if (base_.walltimeExceeded(t0)) {
					break;
				}
			}
		}
		QK_HOST_HIP(hipDeviceSynchronize());
		elapsedSeconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if constexpr (!Sim::isAdvection) {
			base_.outputAfterEvolve();
		}
		base_.elapsedSeconds_ = elapsedSeconds_;
		base_.cellUpdates_ = cellUpdates_;
		base_.computeAfterEvolve(init_sum_cons);
		if constexpr (!Sim::isAdvection) {
			base_.printConservation(init_sum_cons, vol); // reference src/simulation.hpp:959-970
		}
		double const us = 1.0e6 * elapsedSeconds_ / static_cast<double>(cellUpdates_);
		amrex::Print() << "Performance figure-of-merit: " << us << " μs/zone-update [" << 1.0 / us << " Mupdates/s]\n";
		for (int l = 0; l <= finestLevel(); ++l) {
			amrex::Print() << "Zone-updates on level " << l << ": " << cellUpdatesEachLevel_[l] << " (" << level(l).allGrids_.size() << " grids)\n";
			if (distLevels_ != 0) {
				std::vector<int> per(static_cast<size_t>(qkhost::Comm::get().size), 0);
				for (int const o : level(l).owner_) {
					++per[static_cast<size_t>(o)];
				}
				amrex::Print() << "Boxes of level " << l << " per rank:";
				for (int const n : per) {
					amrex::Print() << " " << n;
				}
				amrex::Print() << "\n";
			}
		}
		if (specOverlapped_ + specRolledBack_ > 0) {
			amrex::Print() << "speculative coarse steps: overlapped=" << specOverlapped_ << " rolled_back=" << specRolledBack_ << "\n";
		}
		for (auto const &kv : phaseSeconds_) {
			amrex::Print() << "host phase " << kv.first << ": " << kv.second << " s\n";
		}
	}

	// the refined levels of AMRSimulation::ReadCheckpointFile (reference src/simulation.hpp:2676-2801): BoxArrays from the Header, data read
	// fab by fab; flux registers, interpolation and average-down plans are rebuilt by makeLevel
	void readCheckpointLevels()
	{
		auto const h = quokka::io::ReadCheckpointHeader(base_.restart_chkfile);
		AMREX_ALWAYS_ASSERT(h.finest_level <= max_level);
		tNew_ = h.tNew.at(0);
		base_.tOldLev_ = base_.tNewLev_ = tNew_;
		for (int lev = 0; lev <= max_level && lev < static_cast<int>(h.istep.size()); ++lev) {
			istep[lev] = h.istep[lev];
			dt_[lev] = h.dt[lev];
		}
		for (int lev = 1; lev <= h.finest_level; ++lev) {
			makeLevel(lev, h.grids[lev]);
			Sim &me = level(lev);
			quokka::io::VisMFReadInto(me.state_new_cc_[0], base_.restart_chkfile + "/Level_" + std::to_string(lev) + "/Cell");
			amrex::MultiFab::Copy(me.state_old_cc_[0], me.state_new_cc_[0]);
			me.tOldLev_ = me.tNewLev_ = h.tNew.at(lev);
			me.areInitialConditionsDefined_ = true;
		}
	}

	// volume integral of a conserved component over the composite grid (coarse cells under a finer level are not counted)
	auto compositeSum(int comp) -> double
	{
		double total = 0;
		for (int l = 0; l <= finestLevel(); ++l) {
			auto &S = level(l);
			auto const &g = S.geom[0];
			double const vol = AMREX_D_TERM(g.dx[0], *g.dx[1], *g.dx[2]);
			auto &mf = S.state_new_cc_[0];
			for (int b = 0; b < mf.size(); ++b) {
				auto h = mf.copyToHost(b);
				amrex::Array4<double> a(h.data(), mf.fabbox(b), mf.nComp());
				double s = 0, c = 0;
				amrex::HostFor(mf.validbox(b), [&](int i, int j, int k) {
					if (l < finestLevel()) {
						for (auto const &fb : level(l + 1).allGrids_) {
							if (fb.contains(2 * i, 2 * j, 2 * k)) {
								return;
							}
						}
					}
					double const y = a(i, j, k, comp) - c;
					double const t = s + y;
					c = (t - s) - y;
					s = t;
				});
				total += s * vol;
			}
		}
		return qkhost::Comm::get().allReduceSum(total);
	}

      private:
	// qk.distribute_levels: the coarse-level data a refined level needs and produces, on the FINE level's distribution — what AMReX builds inside
	// FillPatchTwoLevels (the coarse patch under the fine boxes' ghost cells: ParallelCopy + the coarse physical boundary conditions, then interpolated
	// locally), average_down (the coarsened fine MultiFab, then ParallelCopy to the coarse level) and YAFluxRegister (m_cfpatch).  Boxes: this rank's
	// fine boxes coarsened; 3 ghost cells (2 under the fine ghost cells + 1 of interpolation stencil).  quokka_amd/amr_simulation.py CoarseShadow.
	struct Shadow {
		static constexpr int NG = 3;
		Sim &parent;
		qk_geometry gc;
		std::vector<qk_box> boxes; // the fine boxes of ALL ranks, coarsened; owners: the fine level's
		std::vector<int> owner;
		std::vector<amrex::Box> mine;
		std::vector<qk_box> mineQ;
		qk_level *lev = nullptr;
		qk_ghost_plan *bcPlan = nullptr; // the physical-boundary slabs of the coarse patch (the cbc of FillPatchTwoLevels)
		amrex::MultiFab oldS, newS;
		std::unique_ptr<qkhost::PcopyPlan> fromParent, fromParentWhole;
		std::map<int, std::unique_ptr<qkhost::PcopyPlan>> toParent_; // by component count
		typename Sim::BcShellCache bcShells;
		// which version of the parent's states the copies hold (descriptor table of the source array, AMRSimulation::GhostFlag::version)
		void const *srcOld = nullptr, *srcNew = nullptr;
		std::uint64_t verOld = 0, verNew = 0;

		Shadow(Sim &parent_, Sim &child) : parent(parent_), gc(qgeom(parent_.geom[0])), owner(child.owner_)
		{
			int const rank = qkhost::Comm::get().rank;
			int const nc = parent.state_new_cc_[0].nComp();
			std::vector<qk_box> holes;
			for (size_t n = 0; n < child.allBoxes_.size(); ++n) {
				qk_box c{}, h{};
				for (int d = 0; d < 3; ++d) {
					c.lo[d] = (d < AMREX_SPACEDIM) ? child.allBoxes_[n].lo[d] >> 1 : 0;
					c.hi[d] = (d < AMREX_SPACEDIM) ? child.allBoxes_[n].hi[d] >> 1 : 0;
					// ghost-cell interpolation reads the three ghost layers and the outermost valid layer of a shadow box, the reflecting boundary
					// conditions of the ghost layers the three outermost valid layers: the rest is a hole in the plan
					h.lo[d] = (d < AMREX_SPACEDIM) ? c.lo[d] + NG : 0;
					h.hi[d] = (d < AMREX_SPACEDIM) ? c.hi[d] - NG : 0;
				}
				boxes.push_back(c);
				holes.push_back(h);
				if (owner[n] == rank) {
					amrex::Box b;
					for (int d = 0; d < 3; ++d) {
						b.lo[d] = c.lo[d];
						b.hi[d] = c.hi[d];
					}
					mine.push_back(b);
					mineQ.push_back(c);
				}
			}
			qk_box const none{};
			qkhost::check(qk_level_create(qkhost::Runtime::get().ctx, &lev, AMREX_SPACEDIM, static_cast<int>(mineQ.size()), mineQ.empty() ? &none : mineQ.data()), "qk_level_create(shadow)");
			oldS.define(mine, nc, NG);
			newS.define(mine, nc, NG);
			fromParent = std::make_unique<qkhost::PcopyPlan>(gc, parent.allBoxes_, parent.owner_, 0, false, boxes, owner, NG, &holes, nc);
			fromParent->name = "shadow <- parent";
			// (only its physical-boundary slabs are used; told the shadow boxes of every rank because a plan wants at least one box)
			qkhost::check(qk_ghost_plan_create(lev, &bcPlan, &gc, NG, nc, static_cast<int>(boxes.size()), boxes.data(), owner.data(), rank), "qk_ghost_plan_create(shadow)");
		}
		Shadow(Shadow const &) = delete;
		auto operator=(Shadow const &) -> Shadow & = delete;
		~Shadow()
		{
			for (auto &kv : bcShells) {
				(void)hipFree(kv.second.d);
			}
			qk_ghost_plan_destroy(bcPlan);
			qk_level_destroy(lev);
		}
		auto toParent(int ncomp) -> qkhost::PcopyPlan &
		{
			auto &p = toParent_[ncomp];
			if (!p) {
				p = std::make_unique<qkhost::PcopyPlan>(gc, boxes, owner, 0, false, parent.allBoxes_, parent.owner_, 0, nullptr, ncomp);
				p->name = "shadow -> parent (average down)";
			}
			return *p;
		}
		void fill(amrex::MultiFab &dst, amrex::MultiFab const &src, qkhost::PcopyPlan &plan, double time)
		{
			plan(src, dst);
			if (!parent.geom[0].isAllPeriodic()) {
				auto const bcs = parent.boundaryRecords();
				qkhost::check(qk_FillPhysicalBoundary_subset(bcPlan, qkhost::Runtime::get().computeStream(), qkhost::tab(dst), bcs.data(), nullptr, QK_BOXES_ALL),
					      "FillPhysicalBoundary(shadow)");
				parent.customBoundaryConditionsOn(dst, QK_BOXES_ALL, bcPlan, parent.geom[0], bcShells, time);
			}
		}
		// the parent's old / new state under this rank's fine boxes — fetched once per version of the parent's state
		auto ensure(bool old) -> amrex::MultiFab &
		{
			amrex::MultiFab &src = old ? parent.state_old_cc_[0] : parent.state_new_cc_[0];
			amrex::MultiFab &dst = old ? oldS : newS;
			void const *&have = old ? srcOld : srcNew;
			std::uint64_t &ver = old ? verOld : verNew;
			if (have != static_cast<void const *>(src.arrays()) || ver != parent.newStateGhostsFilled_.version) {
				fill(dst, src, *fromParent, old ? parent.tOldLev_ : parent.tNewLev_);
				have = static_cast<void const *>(src.arrays());
				ver = parent.newStateGhostsFilled_.version;
			}
			return dst;
		}
		// every cell of the grown shadow boxes from the parent's new state (a (re)made level interpolates all of its cells)
		auto fillWholeNew() -> amrex::MultiFab &
		{
			if (!fromParentWhole) {
				fromParentWhole = std::make_unique<qkhost::PcopyPlan>(gc, parent.allBoxes_, parent.owner_, 0, false, boxes, owner, NG, nullptr, newS.nComp());
				fromParentWhole->name = "whole shadow <- parent";
			}
			fill(newS, parent.state_new_cc_[0], *fromParentWhole, parent.tNewLev_);
			srcNew = nullptr;
			return newS;
		}
	};
	// the fine part of a distributed flux register (m_cfpatch): increments in the one-cell ghost ring of the shadow boxes, sent to the owners of the coarse cells
	struct RingIncrement {
		amrex::MultiFab inc;
		std::unique_ptr<qkhost::PcopyPlan> toCrse;
	};
	struct Finer {
		std::unique_ptr<Sim> sim;
		qk_interp_plan *interp = nullptr;
		qk_fluxreg *fluxreg = nullptr;
		qk_fluxreg *fluxregRad = nullptr; // the radiation block of the state (expandFluxArrays, reference src/QuokkaSimulation.hpp:1758)
		qk_avgdown_plan *avgdown = nullptr;
		// qk.distribute_levels (quokka_amd/amr.py DistFluxRegister): fluxreg / fluxregRad above are the COARSE part (m_crse_data: lives with the coarse
		// boxes, takes CrseAdd), these the FINE part (lives with the fine boxes, takes FineAdd, is saved / restored around retries); otherwise both
		// names mean the one register
		qk_fluxreg *fluxregFinePart = nullptr, *fluxregRadFinePart = nullptr;
		qk_level *allFine = nullptr; // box metadata of the whole fine level (the coarse part is built against it)
		std::unique_ptr<Shadow> shadow;
		std::unique_ptr<RingIncrement> ring, ringRad;
		[[nodiscard]] auto fineSide() const -> qk_fluxreg * { return fluxregFinePart != nullptr ? fluxregFinePart : fluxreg; }
		[[nodiscard]] auto fineSideRad() const -> qk_fluxreg * { return fluxregRadFinePart != nullptr ? fluxregRadFinePart : fluxregRad; }
		void dropPlans()
		{
			qk_interp_plan_destroy(interp);
			qk_fluxreg_destroy(fluxreg);
			qk_fluxreg_destroy(fluxregRad);
			qk_fluxreg_destroy(fluxregFinePart);
			qk_fluxreg_destroy(fluxregRadFinePart);
			qk_avgdown_plan_destroy(avgdown);
			qk_level_destroy(allFine);
			interp = nullptr;
			fluxreg = fluxregRad = fluxregFinePart = fluxregRadFinePart = nullptr;
			avgdown = nullptr;
			allFine = nullptr;
			ring.reset();
			ringRad.reset();
			shadow.reset();
		}
		// several ranks: the reflux increments of the PARENT level (valid + 1 ghost cell: a register cell may belong to a neighbouring rank's box),
		// folded onto their owners by SumBoundary over a 1-ghost plan of the parent's grids, then added to the parent's state
		struct Fold {
			amrex::MultiFab inc;
			qk_ghost_plan *plan = nullptr;
			qkhost::PeerBuffers peers;
			~Fold() { qk_ghost_plan_destroy(plan); }
		};
		std::unique_ptr<Fold> fold, foldRad;
		~Finer() { dropPlans(); }
	};
	Sim &base_;
	std::vector<std::shared_ptr<Finer>> finerOwned_; // (shared: the snapshot of a speculative coarse step keeps the levels a regrid inside it replaces)
	std::vector<Finer *> finer_; // finer_[l-1] = level l
	// a specialised user hook may have changed the state of any level: none of them may reuse the signal speeds its last stage cached
	void dropSignalsIfHooked(bool hookIsDefault)
	{
		if (!hookIsDefault) {
			base_.newStateGhostsFilled_ = false;
			for (auto *f : finer_) {
				if (f != nullptr) {
					f->sim->dropCachedSignal();
					f->sim->newStateGhostsFilled_ = false;
				}
			}
		}
	}

	static auto qgeom(amrex::Geometry const &g) -> qk_geometry
	{
		qk_geometry q{};
		for (int d = 0; d < 3; ++d) {
			q.domain.lo[d] = g.domain.lo[d];
			q.domain.hi[d] = g.domain.hi[d];
			q.periodic[d] = g.periodic[d];
		}
		q.ndim = AMREX_SPACEDIM;
		return q;
	}

	// AmrMesh::ChopGrids: a level with fewer boxes than ranks is cut until every rank can own one (as far as the blocking factor allows)
	auto chop(int lev, std::vector<amrex::Box> boxes) -> std::vector<amrex::Box>
	{
		int const target = (refineTarget_ > 0) ? refineTarget_ : (distLevels_ != 0 ? qkhost::Comm::get().size : 1);
		if (target <= 1 || static_cast<int>(boxes.size()) >= target || boxes.empty()) {
			return boxes;
		}
		std::array<int, 3> len{};
		for (int d = 0; d < 3; ++d) {
			len[d] = (d < AMREX_SPACEDIM) ? base_.geom[0].domain.length(d) * (1 << lev) : 1;
		}
		return qkhost::chopGrids(std::move(boxes), target, max_grid_size, blocking_factor, len, AMREX_SPACEDIM);
	}

	auto specFor(int lev, std::vector<amrex::Box> const &boxes) -> LevelSpec
	{
		LevelSpec s;
		s.geom = base_.geom[0];
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			s.geom.domain.hi[d] = (base_.geom[0].domain.hi[d] + 1) * (1 << lev) - 1;
			s.geom.dx[d] = base_.geom[0].dx[d] / (1 << lev);
		}
		s.boxes = boxes;
		s.level = lev;
		if (distLevels_ != 0) { // the level's own map: space-filling curve over its boxes, the least loaded ranks (cells of the coarser levels) first
			int const nranks = qkhost::Comm::get().size;
			std::vector<long long> load(static_cast<size_t>(nranks), 0);
			for (int l = 0; l < lev; ++l) {
				Sim &L = level(l);
				for (size_t n = 0; n < L.allGrids_.size(); ++n) {
					load[static_cast<size_t>(L.owner_[n])] += L.allGrids_[n].numPts();
				}
			}
			s.owner = qkhost::distributeSfc(boxes, nranks, load, blocking_factor);
		} else if (multi_) { // a box of level lev >= 1 lives on the rank of the level-0 box that contains it
			for (auto const &b : boxes) {
				int owner = -1;
				for (size_t n = 0; n < base_.allGrids_.size() && owner < 0; ++n) {
					if (base_.allGrids_[n].contains(b.lo[0] >> lev, b.lo[1] >> lev, b.lo[2] >> lev)) {
						owner = base_.owner_[n];
					}
				}
				AMREX_ALWAYS_ASSERT(owner >= 0); // (a fine box without a level-0 ancestor)
				s.owner.push_back(owner);
			}
		}
		return s;
	}

	void linkToParent(Finer &f, int lev)
	{
		f.dropPlans();
		Sim &parent = level(lev - 1);
		Sim &me = *f.sim;
		int const ratio[3] = {2, 2, 2};
		auto gf = qgeom(me.geom[0]);
		auto gc = qgeom(parent.geom[0]);
		if (distLevels_ != 0) {
			linkToParentDistributed(f, parent, me, gf, gc);
			installLevelHooks(f, lev);
			return;
		}
		// several ranks: the plans are told the fine boxes of ALL ranks (a ghost cell under a remote fine box is filled by the fine-fine exchange;
		// a register cell owned by another rank is kept in a ghost cell of a local coarse box)
		int const nAll = multi_ ? static_cast<int>(me.allBoxes_.size()) : 0;
		qk_box const *all = multi_ ? me.allBoxes_.data() : nullptr;
		int const regGhost = multi_ ? 1 : 0;
		qkhost::check(qk_interp_plan_create(parent.levelHandle(), me.levelHandle(), &gf, me.nghost_cc_, ratio, 0, nAll, all, &f.interp), "qk_interp_plan_create");
		qkhost::check(qk_fluxreg_create(parent.levelHandle(), me.levelHandle(), &gc, ratio, Sim::ncompHydro_, nAll, all, regGhost, &f.fluxreg), "qk_fluxreg_create");
		qkhost::check(qk_avgdown_plan_create(parent.levelHandle(), me.levelHandle(), ratio, &f.avgdown), "qk_avgdown_plan_create");
		f.fold.reset();
		f.foldRad.reset();
		auto makeFold = [&](int ncomp) {
			auto fo = std::make_unique<typename Finer::Fold>();
			fo->inc.define(parent.grids_, ncomp, 1);
			qkhost::check(qk_ghost_plan_create(parent.levelHandle(), &fo->plan, &gc, 1, ncomp, static_cast<int>(parent.allBoxes_.size()), parent.allBoxes_.data(),
							   parent.owner_.data(), qkhost::Comm::get().rank),
				      "qk_ghost_plan_create(reflux)");
			fo->peers.build(fo->plan, sizeof(double));
			return fo;
		};
		if (multi_) {
			f.fold = makeFold(Sim::ncompHydro_);
		}
		if constexpr (Physics_Traits<problem_t>::is_radiation_enabled) {
			qkhost::check(qk_fluxreg_create(parent.levelHandle(), me.levelHandle(), &gc, ratio, RadSystem<problem_t>::nvarHyperbolic_, nAll, all, regGhost, &f.fluxregRad),
				      "qk_fluxreg_create");
			if (multi_) {
				f.foldRad = makeFold(RadSystem<problem_t>::nvarHyperbolic_);
			} else {
				qkhost::check(qk_fluxreg_set_state_component(f.fluxregRad, RadSystem<problem_t>::nstartHyperbolic_), "qk_fluxreg_set_state_component");
			}
		}
		installLevelHooks(f, lev);
	}

	// the plans between a level and its parent when both have their own box -> rank map (quokka_amd/amr_simulation.py AmrLevelSim.link_to_parent)
	void linkToParentDistributed(Finer &f, Sim &parent, Sim &me, qk_geometry const &gf, qk_geometry const &gc)
	{
		int const ratio[3] = {2, 2, 2};
		f.shadow = std::make_unique<Shadow>(parent, me);
		Shadow &sh = *f.shadow;
		int const nAll = static_cast<int>(me.allBoxes_.size());
		qk_box const *all = me.allBoxes_.data();
		qkhost::check(qk_interp_plan_create(sh.lev, me.levelHandle(), &gf, me.nghost_cc_, ratio, 0, nAll, all, &f.interp), "qk_interp_plan_create");
		qkhost::check(qk_avgdown_plan_create(sh.lev, me.levelHandle(), ratio, &f.avgdown), "qk_avgdown_plan_create");
		qkhost::check(qk_level_create(qkhost::Runtime::get().ctx, &f.allFine, AMREX_SPACEDIM, nAll, all), "qk_level_create(all fine boxes)");
		auto makeRegister = [&](int ncomp, qk_fluxreg **crsePart, qk_fluxreg **finePart) {
			qkhost::check(qk_fluxreg_create_crse_part(parent.levelHandle(), f.allFine, &gc, ratio, ncomp, crsePart), "qk_fluxreg_create_crse_part");
			qkhost::check(qk_fluxreg_create(sh.lev, me.levelHandle(), &gc, ratio, ncomp, nAll, all, 1, finePart), "qk_fluxreg_create(fine part)");
			auto r = std::make_unique<RingIncrement>();
			r->inc.define(sh.mine, ncomp, 1);
			r->toCrse = std::make_unique<qkhost::PcopyPlan>(gc, sh.boxes, sh.owner, 1, true, parent.allBoxes_, parent.owner_, 0, nullptr, ncomp);
			r->toCrse->name = "register ring -> coarse owners";
			return r;
		};
		f.ring = makeRegister(Sim::ncompHydro_, &f.fluxreg, &f.fluxregFinePart);
		if constexpr (Physics_Traits<problem_t>::is_radiation_enabled) {
			f.ringRad = makeRegister(RadSystem<problem_t>::nvarHyperbolic_, &f.fluxregRad, &f.fluxregRadFinePart);
			qkhost::check(qk_fluxreg_set_state_component(f.fluxregRad, RadSystem<problem_t>::nstartHyperbolic_), "qk_fluxreg_set_state_component");
		}
	}

	void installLevelHooks(Finer &f, int lev)
	{
		Sim &parent = level(lev - 1);
		Sim &me = *f.sim;
		Finer *fp = &f;
		// FillPatchTwoLevels: the ghost cells no fine box covers come from the parent, interpolated in space and time
		me.beforePhysBC_ = [this, fp, lev](amrex::MultiFab &state) { interpFromParent(*fp, lev, state, fp->sim->fillTime_, fp->interp); };
		// incrementFluxRegisters: this level as the fine side of its register and as the coarse side of its child's
		if constexpr (Sim::isAdvection) { // (both RK stages add their fluxes with half the step: AdvectionSimulation.hpp:300-349)
			me.afterStageFluxes_ = [this, lev](std::array<amrex::MultiFab, AMREX_SPACEDIM> &flux, double dtw) { addFluxesToRegisters(lev, flux, dtw); };
		} else {
			me.afterAdvance_ = [this, lev](double dt) { incrementFluxRegisters(lev, dt); };
			me.beforeAttempt_ = [this, lev](int retry) { resetFluxRegistersForAttempt(lev, retry); };
			me.storeFluxRk2_ = true;
			if (lev == 1 && parent.wantsCarriedForm()) {
				parent.setFluxMaskFrom(f.fluxreg); // the base level in the carried form, flux_rk2 on its coarse-fine faces only
			}
			installRadiationHook(lev);
			installRadiationHook(lev - 1);
		}
	}

	// incrementFluxRegisters of the radiation stages (:1818, :1854): weight dt_radiation / 2 for each of the two stages
	void installRadiationHook(int lev)
	{
		if constexpr (Physics_Traits<problem_t>::is_radiation_enabled) {
			level(lev).afterRadStage_ = [this, lev](std::array<amrex::MultiFab, AMREX_SPACEDIM> &flux, double dt_radiation) {
				if (do_reflux == 0) {
					return;
				}
				Sim &S = level(lev);
				qk_array4 *f[3];
				double dx[3];
				for (int d = 0; d < 3; ++d) {
					f[d] = (d < AMREX_SPACEDIM) ? qkhost::tab(flux[d]) : nullptr;
					dx[d] = (d < AMREX_SPACEDIM) ? S.geom[0].dx[d] : 1.0;
				}
				if (lev < finestLevel() && finer_[lev] != nullptr && finer_[lev]->fluxregRad != nullptr) {
					qkhost::check(qk_fluxreg_CrseAdd(finer_[lev]->fluxregRad, nullptr, f, dx, 0.5 * dt_radiation), "qk_fluxreg_CrseAdd(rad)");
				}
				if (lev > 0 && finer_[lev - 1]->fluxregRad != nullptr) {
					qkhost::check(qk_fluxreg_FineAdd(finer_[lev - 1]->fineSideRad(), nullptr, f, dx, 0.5 * dt_radiation), "qk_fluxreg_FineAdd(rad)");
				}
			};
		}
	}

	void interpFromParent(Finer &f, int lev, amrex::MultiFab &state, double time, qk_interp_plan *plan)
	{
		Sim &p = level(lev - 1);
		double const t0 = p.tOldLev_, t1 = p.tNewLev_;
		double const eps = 1.0e-10 * std::max(std::abs(t1 - t0), 1.0e-300);
		auto *fs = qkhost::tab(state);
		bool const atNew = std::abs(time - t1) <= eps || t1 == t0;
		bool const atOld = !atNew && std::abs(time - t0) <= eps;
		// distributed levels: the parent's states as this rank's copy under its fine boxes (same values: the same interpolated bits)
		auto *pn = (f.shadow && !atOld) ? qkhost::tab(f.shadow->ensure(false)) : qkhost::tab(p.state_new_cc_[0]);
		auto *po = (f.shadow && !atNew) ? qkhost::tab(f.shadow->ensure(true)) : qkhost::tab(p.state_old_cc_[0]);
		// debugging aid: QK_DUMP_COARSE_FOR_INTERP="<level> <tmin> <prefix> [skip]" writes the parent's old and new state as the (skip + 1)-th ghost
		// interpolation of that level at time >= tmin reads them (ghost cells included; ancestor scheme: the parent's own arrays)
		if (char const *e = std::getenv("QK_DUMP_COARSE_FOR_INTERP")) {
			int l = -1, skip = 0;
			double tmin = 0;
			char prefix[512];
			static int seen = 0;
			if (!f.shadow && std::sscanf(e, "%d %lf %500s %d", &l, &tmin, prefix, &skip) >= 3 && l == lev && time >= tmin && seen++ == skip) {
				std::ostringstream head;
				head << "time " << std::setprecision(17) << time << " t0 " << t0 << " t1 " << t1;
				qkhost::dumpFabs(prefix, head.str(), {&p.state_old_cc_[0]});
				qkhost::dumpFabs(std::string(prefix) + "_new", head.str(), {&p.state_new_cc_[0]});
			}
		}
		int const nc = Physics_Indices<problem_t>::nvarTotal_cc; // hydro + radiation blocks
		int const hooks = Sim::isAdvection ? 0 : 1;		 // PreInterpState / PostInterpState: the hydro energy (InterpHookNone for the scalar)
		if (std::abs(time - t1) <= eps || t1 == t0) {
			qkhost::check(qk_InterpFromCoarse(plan, nullptr, fs, pn, pn, 1.0, 0.0, nc, amrInterpMethod_, hooks), "qk_InterpFromCoarse");
		} else if (std::abs(time - t0) <= eps) {
			qkhost::check(qk_InterpFromCoarse(plan, nullptr, fs, po, po, 1.0, 0.0, nc, amrInterpMethod_, hooks), "qk_InterpFromCoarse");
		} else {
			qkhost::check(qk_InterpFromCoarse(plan, nullptr, fs, po, pn, (t1 - time) / (t1 - t0), (time - t0) / (t1 - t0), nc, amrInterpMethod_, hooks),
				      "qk_InterpFromCoarse");
		}
	}

	// advanceHydroAtLevelWithRetries: the register this level is the fine side of is saved before the first attempt and copied back at every
	// retry; the one it is the coarse side of is reset (reference src/QuokkaSimulation.hpp:894-900, :919-929) — the substeps of a failed
	// attempt that succeeded have already been added
	void resetFluxRegistersForAttempt(int lev, int retry)
	{
		if (do_reflux == 0) {
			return;
		}
		if (lev > 0 && finer_[lev - 1] && finer_[lev - 1]->fluxreg != nullptr) {
			qk_fluxreg *fine = finer_[lev - 1]->fineSide();
			qkhost::check(retry == 0 ? qk_fluxreg_save(fine, nullptr) : qk_fluxreg_restore(fine, nullptr), "qk_fluxreg_save/restore");
		}
		if (retry > 0 && lev < finestLevel() && finer_[lev] && finer_[lev]->fluxreg != nullptr) {
			resetRegister(*finer_[lev], false);
		}
	}

	void incrementFluxRegisters(int lev, double dt)
	{
		if constexpr (!Sim::isAdvection) {
			addFluxesToRegisters(lev, level(lev).halfFlux(), dt);
		}
	}
	void addFluxesToRegisters(int lev, std::array<amrex::MultiFab, AMREX_SPACEDIM> &flux, double dt)
	{
		if (do_reflux == 0) {
			return;
		}
		Sim &S = level(lev);
		qk_array4 *f[3];
		double dx[3];
		for (int d = 0; d < 3; ++d) {
			f[d] = (d < AMREX_SPACEDIM) ? qkhost::tab(flux[d]) : nullptr;
			dx[d] = (d < AMREX_SPACEDIM) ? S.geom[0].dx[d] : 1.0;
		}
		if (lev < finestLevel()) {
			qkhost::check(qk_fluxreg_CrseAdd(finer_[lev]->fluxreg, nullptr, f, dx, dt), "qk_fluxreg_CrseAdd");
		}
		if (lev > 0) {
			qkhost::check(qk_fluxreg_FineAdd(finer_[lev - 1]->fineSide(), nullptr, f, dx, dt), "qk_fluxreg_FineAdd");
		}
	}
	// both parts of a level's register (one register unless the levels are distributed)
	void resetRegister(Finer &f, bool radiation)
	{
		qk_fluxreg *crse = radiation ? f.fluxregRad : f.fluxreg;
		qk_fluxreg *fine = radiation ? f.fluxregRadFinePart : f.fluxregFinePart;
		qkhost::check(qk_fluxreg_reset(crse, nullptr), "qk_fluxreg_reset");
		if (fine != nullptr) {
			qkhost::check(qk_fluxreg_reset(fine, nullptr), "qk_fluxreg_reset(fine part)");
		}
	}

	void makeLevel(int lev, std::vector<amrex::Box> const &boxes)
	{
		if (std::getenv("QK_AMR_VERBOSE") != nullptr) {
			std::cout << "makeLevel " << lev << ": " << boxes.size() << " boxes";
			for (auto const &b : boxes) {
				std::cout << " [" << b.lo[0] << "," << b.lo[1] << "," << b.lo[2] << ":" << b.hi[0] << "," << b.hi[1] << "," << b.hi[2] << "]";
			}
			std::cout << "\n";
		}
		auto f = std::make_unique<Finer>();
		auto spec = specFor(lev, boxes);
		f->sim = std::make_unique<Sim>(base_.BCs_cc_, spec);
		Sim &me = *f->sim;
		me.inheritSettings(base_); // (the problem sets them on the level-0 object in problem_main)
		me.tOldLev_ = me.tNewLev_ = tNew_;
		Finer *raw = f.get();
		if (lev - 1 < static_cast<int>(finer_.size())) {
			finer_[lev - 1] = raw;
			finerOwned_[lev - 1] = std::move(f);
		} else {
			finer_.push_back(raw);
			finerOwned_.push_back(std::move(f));
		}
		linkToParent(*raw, lev);
		if (lev == 1) {
			if constexpr (Sim::isAdvection) {
				base_.afterStageFluxes_ = [this](std::array<amrex::MultiFab, AMREX_SPACEDIM> &flux, double dtw) { addFluxesToRegisters(0, flux, dtw); };
			} else {
				base_.afterAdvance_ = [this](double dt) { incrementFluxRegisters(0, dt); };
				base_.beforeAttempt_ = [this](int retry) { resetFluxRegistersForAttempt(0, retry); };
			}
		}
	}

	void fillGhosts(int lev, amrex::MultiFab &state, double time)
	{
		Sim &S = level(lev);
		S.fillTime_ = time;
		S.fillBoundaryConditions(state);
		if (&state == &S.state_new_cc_[0] && time == S.tNewLev_) {
			S.newStateGhostsFilled_ = true; // until something writes the new state (every writer below clears the flag)
		}
	}

	// ErrorEst -> buffered tags -> blocking-factor tiles -> boxes of level lev+1.  baseLev: the level whose regrid this is (AmrCore::regrid):
	// the levels above it are rebuilt together, finest first, so the nesting domain of level lev+1 comes from the grids of baseLev
	// (which stay) — level lev is rebuilt afterwards around the new level lev+1.  Default: lev itself (its grids stay).
	auto newGrids(int lev, std::vector<amrex::Box> const *finerBoxes, int baseLev = -1) -> std::vector<amrex::Box>
	{
		int const base = (baseLev < 0) ? lev : baseLev;
		for (int l = 0; l <= lev; ++l) {
			// (not again where the ghost cells are current: level 0 was filled for its children just before regrid(1) runs, and regrid(0) tags two
			// levels that both need it — one whole-level fill per coarse step and a half, 2 % of the step)
			if (!level(l).newStateGhostsFilled_) {
				fillGhosts(l, level(l).state_new_cc_[0], level(l).tNewLev_);
			}
		}
		Sim &S = level(lev);
		amrex::TagBoxArray tags;
		tags.define(S.grids_, 1, 0);
		tags.setVal(amrex::TagBox::CLEAR);
		S.ErrorEst(lev, tags, S.tNewLev_, 0);
		int const tile = blocking_factor / 2;
		auto const &dom = S.geom[0].domain;
		AMREX_ALWAYS_ASSERT(tile >= 4);
		int nt[3];
		for (int d = 0; d < 3; ++d) {
			nt[d] = (d < AMREX_SPACEDIM) ? dom.length(d) / tile : 1;
		}
		std::vector<int> flags(static_cast<size_t>(nt[0]) * nt[1] * nt[2], 0);
		qk_box qd{{dom.lo[0], dom.lo[1], dom.lo[2]}, {dom.hi[0], dom.hi[1], dom.hi[2]}};
		int const per[3] = {S.geom[0].periodic[0], S.geom[0].periodic[1], S.geom[0].periodic[2]};
		qkhost::check(qk_amr_tile_flags_periodic(S.levelHandle(), nullptr, reinterpret_cast<qk_carray4 *>(tags.arrays()), &qd, per, n_error_buf, tile, flags.data()),
			      "qk_amr_tile_flags_periodic");
		qkhost::Comm::get().allReduceMaxInts(flags.data(), flags.size()); // every rank clusters the same global flags
		if (std::getenv("QK_AMR_VERBOSE") != nullptr) {
			unsigned long h = 1469598103934665603UL;
			long nset = 0;
			for (int v : flags) {
				h = (h ^ static_cast<unsigned long>(v)) * 1099511628211UL;
				nset += v;
			}
			std::cout << "tileflags lev " << lev << " base " << base << " istep " << istep[0] << " set " << nset << " hash " << h << "\n";
		}
		auto at = [&](int i, int j, int k) -> int & { return flags[static_cast<size_t>(i) + static_cast<size_t>(nt[0]) * (j + static_cast<size_t>(nt[1]) * k)]; };
		if (finerBoxes != nullptr) { // level lev+2 boxes: their level-lev footprint grown by 2 cells must be refined (proper nesting)
			for (auto const &b : *finerBoxes) { // (through a periodic face the footprint continues on the other side of the domain)
				int a[3], e[3];
				for (int d = 0; d < 3; ++d) {
					bool const per = d < AMREX_SPACEDIM && S.geom[0].periodic[d] != 0;
					a[d] = fdiv(fdiv(b.lo[d], 4) - 2, tile);
					e[d] = fdiv(fdiv(b.hi[d], 4) + 2, tile);
					if (!per) {
						a[d] = std::max(a[d], 0);
						e[d] = std::min(e[d], nt[d] - 1);
					}
				}
				for (int k = a[2]; k <= e[2]; ++k) {
					for (int j = a[1]; j <= e[1]; ++j) {
						for (int i = a[0]; i <= e[0]; ++i) {
							at(((i % nt[0]) + nt[0]) % nt[0], ((j % nt[1]) + nt[1]) % nt[1], ((k % nt[2]) + nt[2]) % nt[2]) = 1;
						}
					}
				}
			}
		}
		std::vector<int> allowed(flags.size(), 1); // the proper-nesting domain, in tiles
		if (base > 0) { // a tile and its 26 neighbours lie on cells of level `base` (refined to this level) or beyond the domain
			int const r = 1 << (lev - base);
			std::vector<char> cov(static_cast<size_t>(nt[0] + 2) * (nt[1] + 2) * (nt[2] + 2), 1);
			auto cv = [&](int i, int j, int k) -> char & {
				return cov[static_cast<size_t>(i + 1) + static_cast<size_t>(nt[0] + 2) * ((j + 1) + static_cast<size_t>(nt[1] + 2) * (k + 1))];
			};
			for (int k = 0; k < nt[2]; ++k) {
				for (int j = 0; j < nt[1]; ++j) {
					for (int i = 0; i < nt[0]; ++i) {
						cv(i, j, k) = 0;
					}
				}
			}
			for (auto const &b : level(base).allGrids_) {
				for (int k = b.lo[2] * r / tile; k <= (b.hi[2] * r + r - 1) / tile; ++k) {
					for (int j = b.lo[1] * r / tile; j <= (b.hi[1] * r + r - 1) / tile; ++j) {
						for (int i = b.lo[0] * r / tile; i <= (b.hi[0] * r + r - 1) / tile; ++i) {
							cv(i, j, k) = 1;
						}
					}
				}
			}
			// Levels base+1 .. lev are rebuilt in the same regrid, each nested in the next coarser one with a margin of one of ITS tiles: seen
			// from level lev the grids of `base` shrink by 2^(lev-base+1) - 2 tiles before the usual one-tile check.  Beyond a periodic face
			// lies the other side of the domain (which level `base` need not cover); beyond a physical one nothing that interpolation would
			// read: only those border tiles stay "covered".
			int const passes = (1 << (lev - base + 1)) - 1;
			std::vector<char> next(cov.size(), 1);
			for (int pass = 0; pass < passes; ++pass) {
				for (int k = -1; k <= nt[2]; ++k) {
					for (int j = -1; j <= nt[1]; ++j) {
						for (int i = -1; i <= nt[0]; ++i) {
							int idx[3] = {i, j, k};
							bool border = false, wraps = true;
							for (int d = 0; d < 3; ++d) {
								if (idx[d] < 0 || idx[d] >= nt[d]) {
									border = true;
									if (d < AMREX_SPACEDIM && S.geom[0].periodic[d] != 0) {
										idx[d] = (idx[d] + nt[d]) % nt[d];
									} else {
										wraps = false;
									}
								}
							}
							if (border && wraps) {
								cv(i, j, k) = cv(idx[0], idx[1], idx[2]);
							}
						}
					}
				}
				next = cov;
				for (int k = 0; k < nt[2]; ++k) {
					for (int j = 0; j < nt[1]; ++j) {
						for (int i = 0; i < nt[0]; ++i) {
							bool ok = true;
							for (int c2 = -1; c2 <= 1 && ok; ++c2) {
								for (int b2 = -1; b2 <= 1 && ok; ++b2) {
									for (int a2 = -1; a2 <= 1 && ok; ++a2) {
										ok = cv(i + a2, j + b2, k + c2) != 0;
									}
								}
							}
							next[static_cast<size_t>(i + 1) + static_cast<size_t>(nt[0] + 2) * ((j + 1) + static_cast<size_t>(nt[1] + 2) * (k + 1))] = ok ? 1 : 0;
						}
					}
				}
				cov = next;
			}
			for (int k = 0; k < nt[2]; ++k) {
				for (int j = 0; j < nt[1]; ++j) {
					for (int i = 0; i < nt[0]; ++i) {
						if (cv(i, j, k) == 0) {
							at(i, j, k) = 0;
							allowed[static_cast<size_t>(i) + static_cast<size_t>(nt[0]) * (j + static_cast<size_t>(nt[1]) * k)] = 0;
						}
					}
				}
			}
		}
		// amr.grid_eff > 0: Berger-Rigoutsos clustering as amrex::AmrMesh::MakeNewGrids (default, the deck's 0.7); <= 0: the round-1 tile rule
		auto cluster = [&](std::vector<int> const &fl, std::vector<int> const &al, int const n3[3], int const origin[3], std::vector<amrex::Box> &boxes) {
			std::vector<qk_box> out(fl.size() + 1);
			int const n = (grid_eff > 0.0) ? qk_amr_cluster_berger_rigoutsos(fl.data(), al.data(), n3, AMREX_SPACEDIM, blocking_factor, max_grid_size, grid_eff, out.data(),
											   static_cast<int>(out.size()))
						       : qk_amr_cluster_tiles(fl.data(), n3, AMREX_SPACEDIM, blocking_factor, max_grid_size, 0, out.data(), static_cast<int>(out.size()));
			AMREX_ALWAYS_ASSERT(n >= 0);
			for (int b = 0; b < n; ++b) {
				amrex::Box bx;
				for (int d = 0; d < 3; ++d) { // (boxes come back in level lev+1 cells relative to the lattice handed in)
					bx.lo[d] = out[b].lo[d] + origin[d];
					bx.hi[d] = out[b].hi[d] + origin[d];
				}
				boxes.push_back(bx);
			}
		};
		std::vector<amrex::Box> boxes;
		if (clusterWithinParent_ == 0) {
			int const origin[3] = {0, 0, 0};
			cluster(flags, allowed, nt, origin, boxes);
			return boxes;
		}
		// every level is clustered inside its level-0 ancestors (one rank each): the level-0 boxes of ALL ranks, in the same order everywhere
		int const s = 1 << lev;
		for (auto const &b0 : base_.allGrids_) {
			int a[3] = {0, 0, 0}, e[3] = {0, 0, 0}, n3[3] = {1, 1, 1}, origin[3] = {0, 0, 0};
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				int const lo = b0.lo[d] * s, hi = b0.hi[d] * s + s - 1;
				AMREX_ALWAYS_ASSERT(lo % tile == 0 && (hi + 1) % tile == 0); // (level-0 boxes are whole tiles: max_grid_size is a multiple of blocking_factor)
				a[d] = lo / tile;
				e[d] = hi / tile;
				n3[d] = e[d] - a[d] + 1;
				origin[d] = 2 * lo;
			}
			std::vector<int> fl, al;
			fl.reserve(static_cast<size_t>(n3[0]) * n3[1] * n3[2]);
			al.reserve(fl.capacity());
			for (int k = a[2]; k <= e[2]; ++k) {
				for (int j = a[1]; j <= e[1]; ++j) {
					for (int i = a[0]; i <= e[0]; ++i) {
						size_t const idx = static_cast<size_t>(i) + static_cast<size_t>(nt[0]) * (j + static_cast<size_t>(nt[1]) * k);
						fl.push_back(flags[idx]);
						al.push_back(allowed[idx]);
					}
				}
			}
			cluster(fl, al, n3, origin, boxes);
		}
		return boxes;
	}
	static auto fdiv(int a, int r) -> int { return (a >= 0) ? a / r : -((-a + r - 1) / r); }

	static auto sameBoxes(std::vector<amrex::Box> const &a, std::vector<amrex::Box> const &b) -> bool
	{
		if (a.size() != b.size()) {
			return false;
		}
		for (size_t n = 0; n < a.size(); ++n) {
			for (int d = 0; d < 3; ++d) {
				if (a[n].lo[d] != b[n].lo[d] || a[n].hi[d] != b[n].hi[d]) {
					return false;
				}
			}
		}
		return true;
	}

	// amrex::AmrCore::regrid(base, time) with RemakeLevel / MakeNewLevelFromCoarse / ClearLevel
	void regrid(int baseLev)
	{
		int const top = std::min(finestLevel() + 1, max_level);
		std::vector<std::vector<amrex::Box>> newBoxes(max_level + 2);
		std::vector<amrex::Box> const *finerB = nullptr;
		for (int lev = top - 1; lev >= baseLev; --lev) {
			if (lev > finestLevel()) {
				continue;
			}
			newBoxes[lev + 1] = newGrids(lev, (lev + 2 <= max_level) ? finerB : nullptr, baseLev);
			finerB = newBoxes[lev + 1].empty() ? nullptr : &newBoxes[lev + 1];
		}
		for (int lev = baseLev + 1; lev <= max_level; ++lev) { // (after the nesting footprints: they follow from the clustered boxes)
			newBoxes[lev] = chop(lev, std::move(newBoxes[lev]));
		}
		for (int lev = baseLev + 1; lev <= max_level; ++lev) {
			auto const &boxes = newBoxes[lev];
			if (boxes.empty()) {
				finer_.resize(lev - 1);
				finerOwned_.resize(lev - 1);
				break;
			}
			bool const existed = lev <= finestLevel();
			if (existed && sameBoxes(level(lev).allGrids_, boxes)) {
				continue;
			}
			std::shared_ptr<Finer> old;
			if (existed) {
				old = finerOwned_[lev - 1];
			}
			makeLevel(lev, boxes);
			Sim &me = level(lev);
			Sim &parent = level(lev - 1);
			Shadow *sh = finer_[lev - 1]->shadow.get();
			int const ratio[3] = {2, 2, 2};
			auto gf = qgeom(me.geom[0]);
			qk_interp_plan *whole = nullptr;
			// distributed levels: the parent's cells under the new boxes arrive in the shadow (with the coarse physical boundaries); the old level's
			// cells from their owners
			amrex::MultiFab *crse = &parent.state_new_cc_[0];
			if (sh != nullptr) {
				crse = &sh->fillWholeNew();
			} else {
				fillGhosts(lev - 1, parent.state_new_cc_[0], parent.tNewLev_);
			}
			qkhost::check(qk_interp_plan_create(sh != nullptr ? sh->lev : parent.levelHandle(), me.levelHandle(), &gf, me.nghost_cc_, ratio, 1, 0, nullptr, &whole),
				      "qk_interp_plan_create(whole)");
			auto *pn = qkhost::tab(*crse);
			qkhost::check(qk_InterpFromCoarse(whole, nullptr, qkhost::tab(me.state_new_cc_[0]), pn, pn, 1.0, 0.0, Physics_Indices<problem_t>::nvarTotal_cc, amrInterpMethod_,
							  Sim::isAdvection ? 0 : 1),
				      "qk_InterpFromCoarse(whole)");
			qk_interp_plan_destroy(whole);
			if (old && sh != nullptr) {
				Sim &o = *old->sim;
				qkhost::PcopyPlan keep(gf, o.allBoxes_, o.owner_, 0, false, me.allBoxes_, me.owner_, 0, nullptr, me.state_new_cc_[0].nComp());
				keep.name = "old level -> new level";
				keep(o.state_new_cc_[0], me.state_new_cc_[0]);
			} else if (old) { // keep the old fine data where the new level still covers it
				Sim &o = *old->sim;
				for (int sb = 0; sb < o.state_new_cc_[0].size(); ++sb) {
					for (int db = 0; db < me.state_new_cc_[0].size(); ++db) {
						int lo[3], hi[3];
						bool ok = true;
						for (int d = 0; d < 3; ++d) {
							lo[d] = std::max(o.grids_[sb].lo[d], me.grids_[db].lo[d]);
							hi[d] = std::min(o.grids_[sb].hi[d], me.grids_[db].hi[d]);
							ok = ok && lo[d] <= hi[d];
						}
						if (!ok) {
							continue;
						}
						auto sa = o.state_new_cc_[0].array(sb);
						auto da = me.state_new_cc_[0].array(db);
						qkhost::check(qk_copy_box(qkhost::Runtime::get().ctx, nullptr, reinterpret_cast<qk_array4 *>(&sa), reinterpret_cast<qk_array4 *>(&da), lo, hi,
									  0, 0, me.state_new_cc_[0].nComp()),
							      "qk_copy_box");
					}
				}
			}
			amrex::MultiFab::Copy(me.state_old_cc_[0], me.state_new_cc_[0]);
			me.tOldLev_ = me.tNewLev_ = parent.tNewLev_;
			QK_HOST_HIP(hipDeviceSynchronize()); // `old` is released below
			// the child of a remade level needs new inter-level plans — unless it is about to be remade (or removed) itself: its OLD grids
			// need not lie inside the new parent (a shrinking hierarchy: the blob of Advection2D leaving through a periodic face)
			if (lev + 1 <= finestLevel() && lev + 1 <= max_level && !newBoxes[lev + 1].empty() && sameBoxes(level(lev + 1).allGrids_, newBoxes[lev + 1])) {
				linkToParent(*finer_[lev], lev + 1);
			}
		}
		for (int lev = baseLev; lev <= finestLevel(); ++lev) {
			level(lev).FixupState(); // reference src/simulation.hpp:1257-1259
			level(lev).newStateGhostsFilled_ = false;
		}
	}

	// flux_reg_[lev+1]->Reflux(state_new_cc_[lev]) (reference src/simulation.hpp:1308).  One rank: straight into the state.  Several ranks: the
	// increments land in a zeroed array with one ghost cell, SumBoundary carries them to their owners (the strips travel in the opposite
	// direction of a ghost fill: receive buffers are sent, send buffers receive), then state(comp0 + n) += increment(n)
	void reflux(Sim &S, qk_fluxreg *reg, typename Finer::Fold *fold, int comp0, qk_fluxreg *finePart = nullptr, RingIncrement *ring = nullptr)
	{
		if (ring != nullptr) { // distributed levels (quokka_amd/amr.py DistFluxRegister.Reflux): the coarse part goes straight into the local state, the
			// fine part travels from the fine boxes' ranks to the owners of the coarse cells (ParallelAdd)
			S.activate();
			hipStream_t const cs = qkhost::Runtime::get().computeStream();
			qkhost::check(qk_fluxreg_Reflux(reg, cs, qkhost::tab(S.state_new_cc_[0])), "qk_fluxreg_Reflux(coarse part)");
			ring->inc.setZeroAsync(cs);
			qkhost::check(qk_fluxreg_Reflux(finePart, cs, qkhost::tab(ring->inc)), "qk_fluxreg_Reflux(fine part)");
			(*ring->toCrse)(ring->inc, S.state_new_cc_[0], 0, comp0, true);
			return;
		}
		if (fold == nullptr) {
			qkhost::check(qk_fluxreg_Reflux(reg, nullptr, qkhost::tab(S.state_new_cc_[0])), "qk_fluxreg_Reflux");
			return;
		}
		S.activate();
		hipStream_t const cs = qkhost::Runtime::get().computeStream();
		fold->inc.setZeroAsync(cs);
		qkhost::check(qk_fluxreg_Reflux(reg, cs, qkhost::tab(fold->inc)), "qk_fluxreg_Reflux");
		auto &pb = fold->peers;
		for (size_t k = 0; k < pb.peer.size(); ++k) {
			qkhost::check(qk_SumBoundary_pack(fold->plan, cs, static_cast<int>(k), qkhost::tab(fold->inc), static_cast<double *>(pb.recv[k])), "SumBoundary_pack");
		}
		qkhost::Comm::get().exchangeBegin(pb.peer, pb.recv, pb.nrecv, pb.send, pb.nsend, sizeof(double), cs);
		qkhost::check(qk_SumBoundary_local(fold->plan, cs, qkhost::tab(fold->inc)), "SumBoundary_local");
		qkhost::Comm::get().exchangeEnd(cs);
		for (size_t k = 0; k < pb.peer.size(); ++k) {
			qkhost::check(qk_SumBoundary_unpack(fold->plan, cs, static_cast<int>(k), qkhost::tab(fold->inc), static_cast<const double *>(pb.send[k])), "SumBoundary_unpack");
		}
		int const nc = fold->inc.nComp();
		for (int b = 0; b < fold->inc.size(); ++b) {
			auto const st = S.state_new_cc_[0].array(b);
			auto const inc = fold->inc.const_array(b);
			amrex::ParallelFor(fold->inc.validbox(b), nc, [=] AMREX_GPU_DEVICE(int i, int j, int k, int n) { st(i, j, k, comp0 + n) += inc(i, j, k, n); });
		}
	}

	void averageDownTo(int crseLev)
	{
		Sim &f = level(crseLev + 1);
		Sim &c = level(crseLev);
		int const nc = Physics_Indices<problem_t>::nvarTotal_cc;
		if (Shadow *sh = finer_[crseLev]->shadow.get()) { // averaged on the fine boxes' ranks, then to the owners of the coarse cells
			qkhost::check(qk_average_down(finer_[crseLev]->avgdown, nullptr, qkhost::tab(f.state_new_cc_[0]), qkhost::tab(sh->newS), 0, nc), "qk_average_down");
			sh->srcNew = nullptr; // (the copy of the parent's new state is gone)
			sh->toParent(nc)(sh->newS, c.state_new_cc_[0]);
			return;
		}
		qkhost::check(qk_average_down(finer_[crseLev]->avgdown, nullptr, qkhost::tab(f.state_new_cc_[0]), qkhost::tab(c.state_new_cc_[0]), 0, nc), "qk_average_down");
	}

	// reference src/simulation.hpp:744-818 with do_subcycle = 1
	void computeTimestep()
	{
		std::vector<double> dt_tmp(finestLevel() + 1);
		for (int l = 0; l <= finestLevel(); ++l) {
			dt_tmp[l] = level(l).computeTimestepAtLevel();
		}
		double dt_0 = dt_tmp[0];
		int n_factor = 1;
		for (int l = 0; l <= finestLevel(); ++l) {
			if (l > 0) {
				n_factor *= 2;
			}
			dt_tmp[l] = std::min(dt_tmp[l], 1.1 * dt_[l]);
			dt_0 = std::min(dt_0, n_factor * dt_tmp[l]);
		}
		double const eps = 1.e-3 * dt_0;
		if (tNew_ + dt_0 > base_.stopTime_ - eps) {
			dt_0 = base_.stopTime_ - tNew_;
		}
		dt_[0] = dt_0;
		for (int l = 1; l <= max_level; ++l) {
			dt_[l] = dt_[l - 1] / 2.0;
		}
	}

	// ------------------------------------------------------------------ the children beside the far boxes (speculative coarse step)
	// Same schedule as quokka_amd/amr_simulation.py (AmrSimulation.overlap_children): stage 2 of the level-0 boxes no child reads runs on a second
	// stream while the children advance; the verdicts — level 0's and those of the children's own level steps — are read once, at the end of the
	// coarse step; a bad one restores states, times, step counters and the grids of a regrid in between and redoes the step in the ordinary order.
	int overlapChildren_ = [] {
		int v = 1;
		amrex::ParmParse("qk").query("overlap_children", v);
		return v;
	}();
	double overlapMaxFineFraction_ = 0.5;
	int forceSpeculationFailureAt_ = [] { // (tests: the rollback path — the verdict of this coarse step is declared bad)
		int v = -1;
		amrex::ParmParse("qk").query("force_speculation_failure_at", v);
		return v;
	}();
	amrex::Long specOverlapped_ = 0, specRolledBack_ = 0;
	static constexpr int specCap_ = 32;
	int64_t *d_specLog_ = nullptr;
	struct SpecEntry {
		Sim *S;
		double dt;
		int slot;
	};
	std::vector<SpecEntry> specEntries_;
	bool deferring_ = false;
	struct SplitCache {
		void const *child = nullptr, *interp = nullptr;
		bool ok = false;
		std::vector<int> near, far;
	} splitCache_;

	// (near, far) boxes of level lev if its children can be advanced beside the second stage of the far boxes.  near: every box the child level
	// reads — the coarse cells under its ghost-cell interpolation (stencil included), the register cells of its flux register, the cells it
	// averages down to.  Required of every interpolation item: the coarse cells it reads lie in the VALID region of its coarse box or beyond a
	// physical boundary (never in ghost cells another box fills: those wait for the far boxes).
	auto speculativeSplit(int lev) -> bool
	{
		if constexpr (Sim::isAdvection) {
			return false;
		} else {
			Sim &L = level(lev);
			if (overlapChildren_ == 0 || lev != 0 || multi_ || distLevels_ != 0 || lev >= finestLevel() || do_reflux == 0 || !L.canSpeculate() || L.grids_.size() < 2) {
				return false;
			}
			amrex::Long fine = 0;
			for (int l = lev + 1; l <= finestLevel(); ++l) {
				fine += level(l).CountCells(0);
			}
			if (static_cast<double>(fine) > overlapMaxFineFraction_ * static_cast<double>(L.CountCells(0))) {
				return false;
			}
			Finer &f = *finer_[lev];
			if (splitCache_.child == &f && splitCache_.interp == f.interp) {
				return splitCache_.ok;
			}
			splitCache_ = SplitCache{};
			splitCache_.child = &f;
			splitCache_.interp = f.interp;
			std::vector<char> isNear(L.grids_.size(), 0);
			bool ok = true;
			auto const &g = L.geom[0];
			for (int n = 0; n < qk_interp_plan_num_items(f.interp); ++n) {
				int fb = 0, cb = 0, lo[3], hi[3];
				qkhost::check(qk_interp_plan_item(f.interp, n, &fb, &cb, lo, hi), "qk_interp_plan_item");
				isNear[static_cast<size_t>(cb)] = 1;
				auto const &v = L.grids_[static_cast<size_t>(cb)];
				for (int d = 0; d < 3; ++d) {
					int const clo = (lo[d] >> 1) - 1, chi = (hi[d] >> 1) + 1; // (arithmetic shift: floor division for the negative ghost indices)
					if (clo < v.lo[d] && !(v.lo[d] == g.domain.lo[d] && g.periodic[d] == 0)) {
						ok = false;
					}
					if (chi > v.hi[d] && !(v.hi[d] == g.domain.hi[d] && g.periodic[d] == 0)) {
						ok = false;
					}
				}
			}
			for (int n = 0; n < qk_fluxreg_num_items(f.fluxreg); ++n) {
				int dir = 0, side = 0, fb = 0, cb = 0, lo[3], hi[3], sh[3];
				qkhost::check(qk_fluxreg_item(f.fluxreg, n, &dir, &side, &fb, &cb, lo, hi, sh), "qk_fluxreg_item");
				isNear[static_cast<size_t>(cb)] = 1;
			}
			for (auto const &fbx : f.sim->allGrids_) { // the cells AverageDown writes
				for (size_t b = 0; b < L.grids_.size(); ++b) {
					bool hit = true;
					for (int d = 0; d < 3; ++d) {
						hit = hit && (fbx.lo[d] >> 1) <= L.grids_[b].hi[d] && (fbx.hi[d] >> 1) >= L.grids_[b].lo[d];
					}
					if (hit) {
						isNear[b] = 1;
					}
				}
			}
			for (size_t b = 0; b < isNear.size(); ++b) {
				(isNear[b] != 0 ? splitCache_.near : splitCache_.far).push_back(static_cast<int>(b));
			}
			splitCache_.ok = ok && !splitCache_.near.empty() && !splitCache_.far.empty();
			if (splitCache_.ok) {
				L.setSpeculativeSplit(splitCache_.near, splitCache_.far);
			}
			return splitCache_.ok;
		}
	}

	// what a rollback must bring back of the levels above `lev`
	struct LevelSnapshot {
		Sim *S;
		amrex::MultiFab data; // state_new_cc_ (the old state is overwritten by the next advance anyway)
		double tOld, tNew;
	};
	struct Snapshot {
		std::vector<std::shared_ptr<Finer>> owned;
		std::vector<Finer *> raw;
		std::vector<int> istep, lastRegrid;
		amrex::Long cu;
		std::vector<amrex::Long> cul;
		std::vector<LevelSnapshot> levels;
	};
	auto snapshotAbove(int lev) -> Snapshot
	{
		Snapshot sn{finerOwned_, finer_, istep, last_regrid_step, cellUpdates_, cellUpdatesEachLevel_, {}};
		for (int l = lev + 1; l <= finestLevel(); ++l) {
			Sim &S = level(l);
			LevelSnapshot ls{&S, amrex::MultiFab(), S.tOldLev_, S.tNewLev_};
			ls.data.define(S.grids_, S.state_new_cc_[0].nComp(), S.state_new_cc_[0].nGrow());
			amrex::MultiFab::Copy(ls.data, S.state_new_cc_[0]);
			sn.levels.push_back(std::move(ls));
		}
		return sn;
	}
	void restoreAbove(int lev, Snapshot &sn)
	{
		QK_HOST_HIP(hipDeviceSynchronize()); // (levels made by a regrid of the discarded attempt are released below)
		finerOwned_ = sn.owned;
		finer_ = sn.raw;
		istep = sn.istep;
		last_regrid_step = sn.lastRegrid;
		cellUpdates_ = sn.cu;
		cellUpdatesEachLevel_ = sn.cul;
		for (auto &ls : sn.levels) {
			Sim &S = *ls.S;
			amrex::MultiFab::Copy(S.state_new_cc_[0], ls.data);
			S.tOldLev_ = ls.tOld;
			S.tNewLev_ = ls.tNew;
			S.oldStateGhostsFilled_ = false;
			S.newStateGhostsFilled_ = false;
			S.clearErrorWords(); // (whatever the discarded attempt flagged: the sticky error word, the cached signal speeds)
		}
		for (int l = lev + 1; l <= finestLevel(); ++l) { // plans a regrid of the discarded attempt may have re-pointed
			linkToParent(*finer_[l - 1], l);
		}
		splitCache_ = SplitCache{};
	}
	// the verdicts of the level steps a speculative coarse step deferred: both stages clean, no error flag, no CFL violation — one device -> host copy
	auto deferredVerdicts() -> bool
	{
		if (specEntries_.empty()) {
			return true;
		}
		std::vector<int64_t> h(static_cast<size_t>(8 * specEntries_.size()));
		QK_HOST_HIP(hipMemcpy(h.data(), d_specLog_, h.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
		bool ok = true;
		std::map<Sim *, std::pair<double, double>> last;
		for (auto const &e : specEntries_) {
			int64_t const *w = &h[static_cast<size_t>(8 * e.slot)];
			int64_t const nbad1 = w[2], nbad2 = w[6];
			int64_t const err = (w[3] | w[7]) & 0xFFFFFFFFLL;
			double sig[2];
			std::memcpy(sig, &w[4], 2 * sizeof(double));
			if (err != 0) {
				amrex::Abort("density is negative in SyncDualEnergy! abort!!");
			}
			if (nbad1 != 0 || nbad2 != 0 || e.dt > 1.1 * e.S->cflLimitFor(sig[0])) {
				ok = false;
				break;
			}
			last[e.S] = {sig[0], sig[1]};
		}
		if (ok) {
			for (auto &kv : last) {
				kv.first->adoptDeferredSignal(kv.second.first, kv.second.second);
			}
		}
		return ok;
	}

	void timeStepWithSubcycling(int lev, double time)
	{
		if (regrid_int > 0 && lev < max_level && istep[lev] > last_regrid_step[lev] && istep[lev] % regrid_int == 0) {
			Phase const ph(*this, "regrid");
			regrid(lev);
			for (int k = lev; k <= finestLevel(); ++k) {
				last_regrid_step[k] = istep[k];
			}
			if constexpr (!Sim::isAdvection) {
				if (lev == 0 && std::getenv("QK_PLOT_AFTER_REGRID") != nullptr) { // (debugging aid: the hierarchy as the regrid left it, afterregrid<step>)
					std::string const keep = base_.plot_file;
					base_.plot_file = "afterregrid";
					base_.istep[0] = istep[0];
					base_.WritePlotFile();
					base_.plot_file = keep;
				}
			}
		}
		Sim &S = level(lev);
		S.tOldLev_ = S.tNewLev_;
		S.tNewLev_ += dt_[lev];
		if (do_reflux != 0 && lev < finestLevel()) {
			resetRegister(*finer_[lev], false);
			if (finer_[lev]->fluxregRad != nullptr) {
				resetRegister(*finer_[lev], true);
			}
		}
		S.newStateGhostsFilled_ = false; // (the advance swaps the states and writes the new one)
		if constexpr (!Sim::isAdvection) {
			if (!deferring_ && speculativeSplit(lev)) {
				Phase const ph(*this, "speculative coarse step");
				Snapshot snap = snapshotAbove(lev);
				if (d_specLog_ == nullptr) {
					QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_specLog_), 8 * specCap_ * sizeof(int64_t)));
				}
				S.advanceLevelBegin(time, dt_[lev]);
				deferring_ = true;
				specEntries_.clear();
				for (int i = 1; i <= 2; ++i) {
					timeStepWithSubcycling(lev + 1, time + (i - 1) * dt_[lev + 1]);
				}
				deferring_ = false;
				bool ok = S.advanceLevelJoin(dt_[lev]);
				ok = deferredVerdicts() && ok;
				if (forceSpeculationFailureAt_ >= 0 && istep[lev] == forceSpeculationFailureAt_) {
					forceSpeculationFailureAt_ = -1;
					ok = false;
				}
				if (ok) {
					++specOverlapped_;
					++istep[lev];
					cellUpdates_ += S.CountCells(0);
					cellUpdatesEachLevel_[lev] += S.CountCells(0);
					if (do_reflux != 0) {
						reflux(S, finer_[lev]->fluxreg, finer_[lev]->fold.get(), 0);
					}
					averageDownTo(lev);
					S.fixupNear();
					S.newStateGhostsFilled_ = false;
					return;
				}
				// the level's step (or a child's) was not clean: the children advanced on a state that will not stand.  Back to the start of the step,
				// then the ordinary order (first-order flux correction / retries, then the children).
				++specRolledBack_;
				S.rollBackSpeculativeStep();
				restoreAbove(lev, snap);
				if (do_reflux != 0 && lev < finestLevel()) {
					qkhost::check(qk_fluxreg_reset(finer_[lev]->fluxreg, nullptr), "qk_fluxreg_reset");
				}
			}
			if (deferring_ && lev > 0 && S.canSpeculate() && static_cast<int>(specEntries_.size()) < specCap_) {
				// a level step inside a speculative coarse step: its verdict is deferred too
				int const slot = static_cast<int>(specEntries_.size());
				S.advanceLevelDeferred(time, dt_[lev], d_specLog_ + 8 * slot);
				specEntries_.push_back({&S, dt_[lev], slot});
			} else {
				Phase const ph(*this, "advance level " + std::to_string(lev));
				if (!S.advanceLevel(time, dt_[lev])) {
					amrex::Abort("QUOKKA FATAL ERROR: Hydro update exceeded max_retries on level " + std::to_string(lev));
				}
			}
		} else {
			Phase const ph(*this, "advance level " + std::to_string(lev));
			if (!S.advanceLevel(time, dt_[lev])) {
				amrex::Abort("QUOKKA FATAL ERROR: Hydro update exceeded max_retries on level " + std::to_string(lev));
			}
		}
		++istep[lev];
		cellUpdates_ += S.CountCells(0);
		cellUpdatesEachLevel_[lev] += S.CountCells(0);
		if (lev < finestLevel()) {
			// the children interpolate their ghost cells from this level's old and new states: both need their own ghost cells (distributed levels:
			// the children fetch this level's VALID cells into their shadows and apply the physical boundaries there — nothing to fill here)
			if (distLevels_ == 0) {
				Phase const ph(*this, "parent ghost fills");
				if (!S.oldStateGhostsFilled_) { // (a direct hydro advance filled them in its first stage, at the same time from the same parent data)
					fillGhosts(lev, S.state_old_cc_[0], S.tOldLev_);
				}
				fillGhosts(lev, S.state_new_cc_[0], S.tNewLev_);
			}
			for (int i = 1; i <= 2; ++i) {
				if (lev < finestLevel()) {
					timeStepWithSubcycling(lev + 1, time + (i - 1) * dt_[lev + 1]);
				}
			}
			if (lev < finestLevel()) {
				Phase const ph(*this, "reflux + average down + fixup");
				if (do_reflux != 0) {
					Finer &fc = *finer_[lev];
					reflux(S, fc.fluxreg, fc.fold.get(), 0, fc.fluxregFinePart, fc.ring.get());
					if (fc.fluxregRad != nullptr) {
						if constexpr (Physics_Traits<problem_t>::is_radiation_enabled) {
							reflux(S, fc.fluxregRad, fc.foldRad.get(), RadSystem<problem_t>::nstartHyperbolic_, fc.fluxregRadFinePart, fc.ringRad.get());
						}
					}
				}
				averageDownTo(lev);
				S.FixupState();
				S.newStateGhostsFilled_ = false;
			}
		}
	}
};

// ------------------------------------------------------------------ QuokkaSimulation entry points: uniform grid or hierarchy
template <typename problem_t> void QuokkaSimulation<problem_t>::setInitialConditions()
{
	int max_level = 0;
	amrex::ParmParse pa("amr");
	pa.query("max_level", max_level);
	if (max_level > 0) {
		amr_ = std::make_shared<AmrDriver<problem_t, QuokkaSimulation<problem_t>>>(*this);
		amr_->setInitialConditions();
	} else {
		AMRSimulation<problem_t>::setInitialConditions();
	}
	outputAfterInitialConditions();
}

// AMRSimulation::WritePlotFile (reference src/simulation.hpp:2294-2336): state_new_cc_ of every level — followed, for problems with a
// face-centred state, by its cell-centre averages 0.5 (f_i + f_i+1) per direction (PlotFileMFAtLevel + AverageFCToCC, :2031-2100) —, no derived
// variables
template <typename problem_t> void QuokkaSimulation<problem_t>::WritePlotFile()
{
	int const nlev = amr_ ? amr_->finestLevel() + 1 : 1;
	std::vector<amrex::MultiFab const *> mf;
	std::vector<amrex::MultiFab> combined; // (kept alive until the file is written)
	std::vector<amrex::Geometry> geoms;
	std::vector<int> steps;
	std::vector<quokka::io::Distribution> dists; // several ranks: every rank writes its own fabs, rank 0 the headers
	std::vector<std::string> names = this->componentNames_cc_;
	constexpr int nfc = qkhost::hasFaceState<problem_t>() ? Physics_Indices<problem_t>::nvarTotal_fc : 0;
	if constexpr (nfc > 0) {
		combined.resize(nlev);
		for (auto const &n : componentNames_fc()) {
			names.push_back(n);
		}
	}
	for (int l = 0; l < nlev; ++l) {
		auto &S = amr_ ? amr_->level(l) : *this;
		if constexpr (nfc > 0) {
			int const ncc = S.state_new_cc_[0].nComp();
			constexpr int per = Physics_Indices<problem_t>::nvarPerDim_fc;
			combined[l].define(S.grids_, ncc + nfc, 0);
			for (int b = 0; b < combined[l].size(); ++b) {
				auto const out = combined[l].array(b);
				auto const cc = S.state_new_cc_[0].const_array(b);
				amrex::ParallelFor(combined[l].validbox(b), ncc, [=] AMREX_GPU_DEVICE(int i, int j, int k, int n) { out(i, j, k, n) = cc(i, j, k, n); });
				for (int idim = 0; idim < AMREX_SPACEDIM; ++idim) {
					auto const fc = S.state_new_fc_[0][idim].const_array(b);
					int const di = (idim == 0) ? 1 : 0, dj = (idim == 1) ? 1 : 0, dk = (idim == 2) ? 1 : 0;
					int const dst = ncc + idim * per;
					amrex::ParallelFor(combined[l].validbox(b), per, [=] AMREX_GPU_DEVICE(int i, int j, int k, int n) {
						out(i, j, k, dst + n) = 0.5 * (fc(i, j, k, n) + fc(i + di, j + dj, k + dk, n));
					});
				}
			}
			mf.push_back(&combined[l]);
		} else {
			mf.push_back(&S.state_new_cc_[0]);
		}
		geoms.push_back(S.geom[0]);
		steps.push_back(amr_ ? amr_->istep[l] : istep[0]);
		dists.push_back({S.allGrids_, S.owner_});
	}
	std::string const name = quokka::io::Concatenate(this->plot_file, istep[0], 5);
	amrex::Print() << "Writing plotfile " << name << "\n";
	QK_HOST_HIP(hipDeviceSynchronize());
	quokka::io::WriteMultiLevelPlotfile(name, nlev, mf, names, geoms, tNew_[0], steps, 2, &dists);
	if (qkhost::Comm::get().rank == 0) {
		quokka::io::WriteMetadataFile(name + "/metadata.yaml");
	}
}

// AMRSimulation::WriteCheckpointFile (reference src/simulation.hpp:2564-2666)
template <typename problem_t> void QuokkaSimulation<problem_t>::WriteCheckpointFile()
{
	quokka::io::CheckpointHeader h;
	h.finest_level = amr_ ? amr_->finestLevel() : 0;
	int const nmax = amr_ ? amr_->max_level + 1 : 1; // istep, dt_, tNew_ have one entry per level that may exist
	std::vector<amrex::MultiFab const *> state;
	std::vector<std::array<amrex::MultiFab const *, AMREX_SPACEDIM>> faces; // Level_<l>/Face_x|y|z of problems with a face-centred state
	std::vector<quokka::io::Distribution> dists;
	for (int l = 0; l < nmax; ++l) {
		bool const live = l <= h.finest_level;
		h.istep.push_back(amr_ ? amr_->istep[l] : istep[0]);
		h.dt.push_back(amr_ ? amr_->dt_[l] : dt_[0]);
		h.tNew.push_back(live ? (amr_ ? amr_->level(l).tNewLev_ : tNew_[0]) : 0.0);
		if (live) {
			auto &S = amr_ ? amr_->level(l) : *this;
			h.grids.push_back(S.allGrids_); // (the boxes of the whole level: with several ranks grids_ holds the local ones)
			dists.push_back({S.allGrids_, S.owner_});
			state.push_back(&S.state_new_cc_[0]);
			if constexpr (qkhost::hasFaceState<problem_t>()) {
				std::array<amrex::MultiFab const *, AMREX_SPACEDIM> f{};
				for (int d = 0; d < AMREX_SPACEDIM; ++d) {
					f[d] = &S.state_new_fc_[0][d];
				}
				faces.push_back(f);
			}
		}
	}
	std::string const name = quokka::io::Concatenate(this->chk_file, istep[0], 5);
	amrex::Print() << "Writing checkpoint " << name << "\n";
	QK_HOST_HIP(hipDeviceSynchronize());
	quokka::io::WriteCheckpointFile(name, h, state, faces, &dists);
}

template <typename problem_t>
template <typename F>
auto QuokkaSimulation<problem_t>::computeAxisAlignedProfile(const int axis, F const &user_f) -> amrex::Gpu::HostVector<amrex::Real>
{
	int const finest = amr_ ? amr_->finestLevel() : 0;
	std::vector<amrex::MultiFab> q(finest + 1);
	for (int lev = 0; lev <= finest; ++lev) {
		auto &S = (lev == 0) ? *this : amr_->level(lev);
		q[lev].define(S.grids_, 1, 0);
		for (int b = 0; b < q[lev].size(); ++b) {
			auto const state = S.state_new_cc_[0].const_array(b);
			auto const result = q[lev].array(b);
			amrex::ParallelFor(q[lev].validbox(b), [=] AMREX_GPU_DEVICE(int i, int j, int k) { result(i, j, k) = user_f(i, j, k, state); });
		}
	}
	for (int crse = finest - 1; crse >= 0; --crse) {
		amr_->averageDownField(crse, q[crse + 1], q[crse]);
	}
	// amrex::sumToLine over the level-0 boxes of this rank, then over ranks (a diagnostic: host reduction)
	amrex::Box const domain = this->geom[0].Domain();
	amrex::Gpu::HostVector<amrex::Real> profile(static_cast<size_t>(domain.length(axis)), 0.0);
	QK_HOST_HIP(hipDeviceSynchronize());
	for (int b = 0; b < q[0].size(); ++b) {
		auto h = q[0].copyToHost(b);
		amrex::Array4<amrex::Real> a(h.data(), q[0].fabbox(b), 1);
		amrex::HostFor(q[0].validbox(b), [&](int i, int j, int k) {
			int const idx[3] = {i, j, k};
			profile[static_cast<size_t>(idx[axis] - domain.lo[axis])] += a(i, j, k);
		});
	}
	for (double &bin : profile) {
		bin = qkhost::Comm::get().allReduceSum(bin);
	}
	amrex::Long const numCells = domain.numPts() / domain.length(axis);
	for (double &bin : profile) {
		bin /= static_cast<amrex::Real>(numCells);
	}
	return profile;
}

template <typename problem_t> void QuokkaSimulation<problem_t>::evolve()
{
	if (this->doPoissonSolve_ != 0) {
		amrex::Abort("doPoissonSolve_ = 1: self-gravity (the MLMG Poisson solve of reference src/simulation.hpp:1014-1095) is not built in quokka_amd/host");
	}
	if (amr_) {
		amr_->evolve();
	} else {
		evolveSingleLevel();
	}
	qkDumpState(*this); // test hook `qk.dump_state=<file>` (no-op without it): also for problem files that are compiled unchanged
}

#endif // QK_HOST_QUOKKA_AMR_HPP_
