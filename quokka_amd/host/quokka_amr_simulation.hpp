// quokka_amr_simulation.hpp — part 3 of the C++17 host mirror: the box -> rank map, peer buffers, SimulationData, LevelSpec and AMRSimulation<problem_t>
//   (reference src/simulation.hpp): deck parsing, geometry, the level's arrays, fillBoundaryConditions (level-0 ghost fill + the problem's
//   setCustomBoundaryConditions as a kernel), output, the hooks a problem specialises.
#ifndef QK_HOST_QUOKKA_AMR_SIMULATION_HPP_
#define QK_HOST_QUOKKA_AMR_SIMULATION_HPP_

#include <unordered_map>
#include <variant>

#include "compat/particles_decl.hpp"
#include "quokka_rad_system.hpp"

namespace qkhost
{
// Locality-preserving box -> rank map (the role of AMReX's SFC DistributionMapping; the same rule as quokka_amd/simulation.py
// distribute_boxes): the box lattice nb[0] x nb[1] x nb[2] is cut into `nranks` bricks by repeatedly halving its longest axis
// (2 x 2 x 2 bricks for 8 ranks); lattices that cannot be cut that way fall back to contiguous blocks of boxes.
inline auto distributeBoxes(int const nb[3], int nranks) -> std::vector<int>
{
	int const nboxes = nb[0] * nb[1] * nb[2];
	std::vector<int> owner(static_cast<size_t>(nboxes), 0);
	if (nranks <= 1) {
		return owner;
	}
	int parts[3] = {1, 1, 1};
	int r = nranks;
	while (r > 1) {
		int d = 0;
		for (int a = 1; a < 3; ++a) {
			if (static_cast<double>(nb[a]) / parts[a] > static_cast<double>(nb[d]) / parts[d]) {
				d = a;
			}
		}
		if (r % 2 != 0 || nb[d] / (parts[d] * 2) < 1) {
			break;
		}
		parts[d] *= 2;
		r /= 2;
	}
	if (parts[0] * parts[1] * parts[2] != nranks) {
		int const per = (nboxes + nranks - 1) / nranks;
		for (int i = 0; i < nboxes; ++i) {
			owner[i] = std::min(i / per, nranks - 1);
		}
		return owner;
	}
	int n = 0;
	for (int kb = 0; kb < nb[2]; ++kb) {
		for (int jb = 0; jb < nb[1]; ++jb) {
			for (int ib = 0; ib < nb[0]; ++ib) {
				int const idx[3] = {ib, jb, kb};
				int p[3];
				for (int d = 0; d < 3; ++d) {
					p[d] = std::min(idx[d] * parts[d] / nb[d], parts[d] - 1);
				}
				owner[n++] = p[0] + parts[0] * (p[1] + parts[1] * p[2]);
			}
		}
	}
	return owner;
}

// rank = Morton index of the box mod nranks (quokka_amd/simulation.py distribute_boxes_interleaved): every neighbourhood of the box lattice is
// spread over all ranks
inline auto distributeBoxesInterleaved(int const nb[3], int nranks) -> std::vector<int>
{
	std::vector<int> owner;
	for (int kb = 0; kb < nb[2]; ++kb) {
		for (int jb = 0; jb < nb[1]; ++jb) {
			for (int ib = 0; ib < nb[0]; ++ib) {
				unsigned m = 0;
				for (int bit = 0; bit < 10; ++bit) {
					m |= ((static_cast<unsigned>(ib) >> bit) & 1U) << (3 * bit) | ((static_cast<unsigned>(jb) >> bit) & 1U) << (3 * bit + 1) |
					     ((static_cast<unsigned>(kb) >> bit) & 1U) << (3 * bit + 2);
				}
				owner.push_back(static_cast<int>(m % static_cast<unsigned>(nranks)));
			}
		}
	}
	return owner;
}

// device send / receive buffers for the peers of a ghost plan (qk_ghost_plan_peer: rank and strip sizes in elements)
struct PeerBuffers {
	std::vector<int> peer;
	std::vector<void *> send, recv;
	std::vector<int64_t> nsend, nrecv;
	void build(qk_ghost_plan *plan, size_t elemBytes)
	{
		int const np = qk_ghost_plan_num_peers(plan);
		for (int k = 0; k < np; ++k) {
			int r = 0;
			int64_t ns = 0, nr = 0;
			check(qk_ghost_plan_peer(plan, k, &r, &ns, &nr), "qk_ghost_plan_peer");
			void *s = nullptr, *rv = nullptr;
			QK_HOST_HIP(hipMalloc(&s, std::max<size_t>(static_cast<size_t>(ns) * elemBytes, 8)));
			QK_HOST_HIP(hipMalloc(&rv, std::max<size_t>(static_cast<size_t>(nr) * elemBytes, 8)));
			peer.push_back(r);
			send.push_back(s);
			recv.push_back(rv);
			nsend.push_back(ns);
			nrecv.push_back(nr);
		}
	}
};
} // namespace qkhost

namespace qkhost
{
// debugging aid (QK_DUMP_BEFORE_SOURCE, QK_DUMP_COARSE_FOR_INTERP): every fab of the arrays, ghost cells included, to <prefix>.rank<r>.bin (doubles, array
// after array); <prefix>.rank<r>.txt: a header line ending in the component count of the FIRST array, then the fab boxes of the first array.
// profiles/tools/compare_source_inputs.py reads two such dumps and lists the cells that differ.
inline void dumpFabs(std::string const &prefix, std::string const &header, std::vector<amrex::MultiFab const *> const &arrays)
{
	std::string const base = prefix + ".rank" + std::to_string(Comm::get().rank);
	std::ofstream meta(base + ".txt"), data(base + ".bin", std::ios::binary);
	meta << header << " ncomp " << arrays.front()->nComp() << "\n";
	QK_HOST_HIP(hipDeviceSynchronize());
	for (auto const *mf : arrays) {
		for (int b = 0; b < mf->size(); ++b) {
			if (mf == arrays.front()) {
				auto const fb = mf->fabbox(b);
				meta << fb.lo[0] << " " << fb.lo[1] << " " << fb.lo[2] << " " << fb.hi[0] << " " << fb.hi[1] << " " << fb.hi[2] << "\n";
			}
			auto const h = mf->copyToHost(b);
			data.write(reinterpret_cast<char const *>(h.data()), static_cast<std::streamsize>(sizeof(double) * h.size()));
		}
	}
}
} // namespace qkhost

// per-problem user data a problem may specialise (reference src/simulation.hpp: SimulationData<problem_t> userData_)
template <typename problem_t> struct SimulationData {
};

template <typename problem_t, typename SimT> class AmrDriver; // quokka_amr.hpp
template <typename problem_t> class AMRSimulation;

namespace qkhost
{
// one slab of ghost cells beyond a non-periodic face of the domain: box index into the array's descriptor table + the cells
struct BcShell {
	int box;
	int lo[3];
	int hi[3];
};
// set by the DEFAULT AMRSimulation<problem_t>::setCustomBoundaryConditions (a problem that specialises the hook never sets it): after the first
// fill the host knows the hook does nothing and stops launching it
static __device__ int g_defaultCustomBcRan = 0;
// setCustomBoundaryConditions on every cell of every slab of a fill: ONE launch per fill (blockIdx.y: the slab), threads over ghost cells only
template <typename problem_t>
__global__ void customBcKernel(const BcShell *shells, const amrex::Array4<amrex::Real> *tab, amrex::GeometryData geom, amrex::Real time, const amrex::BCRec *bcr, int ncomp)
{
	const BcShell sh = shells[blockIdx.y];
	const int nx = sh.hi[0] - sh.lo[0] + 1, ny = sh.hi[1] - sh.lo[1] + 1, nz = sh.hi[2] - sh.lo[2] + 1;
	const amrex::Long n = static_cast<amrex::Long>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (n >= static_cast<amrex::Long>(nx) * ny * nz) {
		return;
	}
	const int k = static_cast<int>(n / (static_cast<amrex::Long>(nx) * ny));
	const int r = static_cast<int>(n - static_cast<amrex::Long>(k) * nx * ny);
	const int j = r / nx;
	const amrex::IntVect iv(sh.lo[0] + (r - j * nx), sh.lo[1] + j, sh.lo[2] + k);
	AMRSimulation<problem_t>::setCustomBoundaryConditions(iv, tab[sh.box], 0, ncomp, geom, time, bcr, 0, 0);
}
} // namespace qkhost

// one refinement level handed to a simulation object by the AMR driver (quokka_amr.hpp): geometry of that level and its boxes
struct LevelSpec {
	amrex::Geometry geom;
	std::vector<amrex::Box> boxes;
	int level = 0;
	std::vector<int> owner; // rank of every box (several ranks: a refined box lives on the rank of its level-0 ancestor); empty: all on rank 0
};

// ---------------------------------------------------------------------------------------------------------------
template <typename problem_t> class AMRSimulation
{
      public:
	// public data members (reference src/simulation.hpp:144-173)
	amrex::Real maxDt_ = std::numeric_limits<double>::max();
	amrex::Real initDt_ = std::numeric_limits<double>::max();
	amrex::Real constantDt_ = 0.0;
	amrex::Real stopTime_ = 1.0;
	amrex::Real cflNumber_ = 0.3;
	amrex::Long maxTimesteps_ = 10000;
	int plotfileInterval_ = -1;   // -1 == no output
	int checkpointInterval_ = -1; // -1 == no output
	std::string plot_file{"plt"}; // plotfile prefix
	std::string chk_file{"chk"};  // checkpoint prefix
	std::string restart_chkfile;  // `restartfile = <checkpoint directory>`
	amrex::Vector<std::string> componentNames_cc_;
	amrex::Real densityFloor_ = 0.0;
	amrex::Real tempFloor_ = 0.0;
	// One simulation object holds ONE level; whatever index a problem's hook uses — state_new_cc_[lev] inside ErrorEst(lev, ...), geom[lev],
	// tNew_[lev] — addresses this object's level.  (geom, tNew_, dt_ and istep were one-element vectors until round 3: Advection2D's ErrorEst
	// reads geom[lev].CellSizeArray() on the refined levels — out of bounds, a different cell size in one tagging out of fifty.)
	template <typename T> struct ThisLevel {
		T item{};
		auto operator[](int /*lev*/) -> T & { return item; }
		auto operator[](int /*lev*/) const -> T const & { return item; }
		auto at(int /*lev*/) -> T & { return item; }
		auto at(int /*lev*/) const -> T const & { return item; }
		[[nodiscard]] auto size() const -> int { return 1; }
	};
	ThisLevel<amrex::Real> tNew_{0.0};
	ThisLevel<amrex::Real> dt_{1.e100};
	ThisLevel<int> istep{0};
	amrex::Long cellUpdates_ = 0;
	int nghost_cc_ = 4;
	bool areInitialConditionsDefined_ = false;

	ThisLevel<amrex::Geometry> geom;
	std::vector<amrex::Box> grids_; // level-0 BoxArray
	amrex::Vector<amrex::BCRec> BCs_cc_;
	// One simulation object holds ONE level (the AMR driver of quokka_amr.hpp owns one object per level): whatever level index a
	// problem's hook uses (state_new_cc_[lev] inside ErrorEst(lev, ...), geom[lev]) addresses this object's level.
	ThisLevel<amrex::MultiFab> state_new_cc_, state_old_cc_;
	[[nodiscard]] auto boxArray(int /*lev*/ = 0) const -> std::vector<amrex::Box> const & { return grids_; }
	[[nodiscard]] auto DistributionMap(int /*lev*/ = 0) const -> amrex::DistributionMapping { return {}; }
	[[nodiscard]] auto finestLevel() const -> int { return 0; }
	// self-gravity (reference src/simulation.hpp:169,392,1014-1095: a Poisson solve with amrex::MLMG per coarse step): NOT built — the members exist
	// so that the problems that set them compile; evolve() refuses a run that asks for the solve
	int doPoissonSolve_ = 0;
	amrex::Real Gconst_ = C::Gconst;
	amrex::Vector<amrex::MultiFab> phi;				   // the potential of that solve: never filled
	std::unique_ptr<quokka::CICParticleContainer> CICParticles; // never created (compat/particles_decl.hpp)
	int finest_level = 0;			     // amrex::AmrMesh::finest_level as a problem's loop over levels reads it: this object's one level
	ThisLevel<amrex::IntVect> ref_ratio{amrex::IntVect(2, 2, 2)};
	// values a problem keeps across restarts (reference src/simulation.hpp:103,175; written to / read from metadata.yaml there — kept in memory here)
	using variant_t = std::variant<amrex::Real, std::string>;
	std::unordered_map<std::string, variant_t> simulationMetadata_;
	amrex::Vector<std::string> derivedNames_; // deck: derived_vars (reference src/simulation.hpp:367,607)

	// integral of user_f(i, j, k, state) over the volume (reference src/simulation.hpp:1966-1991)
	template <typename F> auto computeVolumeIntegral(F const &user_f) -> amrex::Real
	{
		amrex::MultiFab q;
		q.define(grids_, 1, 0);
		for (int b = 0; b < q.size(); ++b) {
			auto const state = state_new_cc_[0].const_array(b);
			auto const result = q.array(b);
			amrex::ParallelFor(q.validbox(b), [=] AMREX_GPU_DEVICE(int i, int j, int k) { result(i, j, k) = user_f(i, j, k, state); });
		}
		amrex::Gpu::streamSynchronize();
		std::vector<amrex::MultiFab const *> const v{&q};
		return amrex::volumeWeightedSum(v, 0, geom, ref_ratio);
	}
	// projection of user_f(i, j, k, state) dx[dir] along `dir` onto the plane of the domain (reference src/simulation.hpp:2394-2450): a host
	// reduction (diagnostics, not on the timed path), summed / minimised over ranks
	template <typename ReduceOp, typename F> auto computePlaneProjection(F const &user_f, const int dir) const -> amrex::BaseFab<amrex::Real>
	{
		amrex::MultiFab q;
		q.define(grids_, 1, 0);
		for (int b = 0; b < q.size(); ++b) {
			auto const state = state_new_cc_[0].const_array(b);
			auto const result = q.array(b);
			amrex::ParallelFor(q.validbox(b), [=] AMREX_GPU_DEVICE(int i, int j, int k) { result(i, j, k) = user_f(i, j, k, state); });
		}
		amrex::Gpu::streamSynchronize();
		amrex::Box plane = geom[0].Domain();
		plane.lo[dir] = 0;
		plane.hi[dir] = 0;
		constexpr bool isSum = std::is_same<ReduceOp, amrex::ReduceOpSum>::value;
		amrex::BaseFab<amrex::Real> proj(plane, 1);
		auto P = proj.array();
		double const init = isSum ? 0.0 : std::numeric_limits<double>::max();
		amrex::HostFor(plane, [&](int i, int j, int k) { P(i, j, k) = init; });
		double const dxdir = geom[0].CellSize(dir);
		for (int b = 0; b < q.size(); ++b) {
			auto h = q.copyToHost(b);
			amrex::Array4<amrex::Real> a(h.data(), q.fabbox(b), 1);
			amrex::HostFor(q.validbox(b), [&](int i, int j, int k) {
				int idx[3] = {i, j, k};
				idx[dir] = 0;
				double const v = dxdir * a(i, j, k);
				double &dst = P(idx[0], idx[1], idx[2]);
				dst = isSum ? dst + v : std::min(dst, v);
			});
		}
		for (amrex::Long n = 0; n < proj.size(); ++n) {
			double &v = proj.dataPtr()[n];
			v = isSum ? qkhost::Comm::get().allReduceSum(v) : qkhost::Comm::get().allReduceMin(v);
		}
		return proj;
	}
	[[nodiscard]] auto Geom(int /*lev*/ = 0) const -> amrex::Geometry const & { return geom[0]; }
	[[nodiscard]] auto Geom(int /*lev*/ = 0) -> amrex::Geometry & { return geom[0]; }
	SimulationData<problem_t> userData_;
	static constexpr int nvarTotal_cc_ = Physics_Indices<problem_t>::nvarTotal_cc;

	explicit AMRSimulation(amrex::Vector<amrex::BCRec> &BCs_cc) : BCs_cc_(BCs_cc) { initialize(nullptr); }
	AMRSimulation(amrex::Vector<amrex::BCRec> &BCs_cc, amrex::Vector<amrex::BCRec> &BCs_fc) : BCs_cc_(BCs_cc), BCs_fc_(BCs_fc) { initialize(nullptr); }
	AMRSimulation(amrex::Vector<amrex::BCRec> &BCs_cc, LevelSpec const &spec) : BCs_cc_(BCs_cc) { initialize(&spec); }
	virtual ~AMRSimulation()
	{
		for (auto &kv : bcShells_) {
			(void)hipFree(kv.second.d);
		}
		(void)hipFree(d_bcrec_);
		if (plan_ != nullptr) {
			qk_ghost_plan_destroy(plan_);
		}
		if (myLev_ != nullptr) {
			if (qkhost::Runtime::get().lev == myLev_) {
				qkhost::Runtime::get().lev = nullptr;
			}
			qk_level_destroy(myLev_);
		}
	}
	struct BcShellList {
		qkhost::BcShell *d = nullptr;
		int count = 0;
		amrex::Long most = 0;
	};
	using BcShellCache = std::map<std::pair<int, int>, BcShellList>;
	// the static operators (HydroSystem<problem_t>::..., RadSystem<problem_t>::...) act on the active level
	void activate() const { qkhost::Runtime::get().lev = myLev_; }
	[[nodiscard]] auto levelHandle() const -> qk_level * { return myLev_; }
	int amrLevel_ = 0;
	// called between FillBoundary and the physical boundaries: the AMR driver interpolates the uncovered ghost cells here
	std::function<void(amrex::MultiFab &)> beforePhysBC_;

	// device hook a problem may specialise (reference src/simulation.hpp:1550-1561); host mode: evaluated on host staging data
	AMREX_GPU_DEVICE static void setCustomBoundaryConditions(const amrex::IntVect & /*iv*/, amrex::Array4<amrex::Real> const & /*dest*/, int /*dcomp*/, int /*numcomp*/,
						amrex::GeometryData const & /*geom*/, amrex::Real /*time*/, const amrex::BCRec * /*bcr*/, int /*bcomp*/,
						int /*orig_comp*/)
	{
		qkhost::g_defaultCustomBcRan = 1; // (see customBoundaryConditionsOnDevice)
	}

	void initialize(LevelSpec const *spec)
	{
		readParameters();
		auto &comm = qkhost::Comm::get();
		comm.init(); // one process per GPU: selects this rank's device (reference src/main.cpp:22-46)
		auto &rt = qkhost::Runtime::get();
		if (rt.ctx == nullptr) {
			int dev = 0;
			QK_HOST_HIP(hipGetDevice(&dev));
			qkhost::check(qk_ctx_create(&rt.ctx, dev), "qk_ctx_create");
		}
		auto &g = geom[0];
		grids_.clear();
		int nb[3] = {1, 1, 1};
		if (spec != nullptr) {
			g = spec->geom;
			grids_ = spec->boxes;
			amrLevel_ = spec->level;
		} else {
			// geometry + BoxArray from the deck (amrex.n_cell, geometry.*, amr.max_grid_size)
			amrex::ParmParse pg("geometry");
			amrex::ParmParse pa("amr");
			std::vector<double> plo{0, 0, 0}, phi{1, 1, 1};
			std::vector<int> per{0, 0, 0}, ncell{32, 32, 32}, mgs;
			pg.queryarr("prob_lo", plo);
			pg.queryarr("prob_hi", phi);
			pg.queryarr("is_periodic", per);
			pa.queryarr("n_cell", ncell);
			if (!pa.queryarr("max_grid_size", mgs) || mgs.empty()) {
				mgs = {128};
			}
			while (mgs.size() < 3) {
				mgs.push_back(mgs.back());
			}
			for (int d = 0; d < 3; ++d) {
				bool const active = d < AMREX_SPACEDIM;
				g.domain.lo[d] = 0;
				g.domain.hi[d] = active ? ncell[d] - 1 : 0;
				g.periodic[d] = active ? per[d] : 0;
				if (active) {
					g.prob_lo[d] = plo[d];
					g.prob_hi[d] = phi[d];
					g.dx[d] = (phi[d] - plo[d]) / ncell[d];
				}
			}
			for (int d = 0; d < 3; ++d) {
				nb[d] = (d < AMREX_SPACEDIM) ? (g.domain.length(d) + mgs[d] - 1) / mgs[d] : 1;
			}
			for (int kb = 0; kb < nb[2]; ++kb) {
				for (int jb = 0; jb < nb[1]; ++jb) {
					for (int ib = 0; ib < nb[0]; ++ib) {
						int const idx[3] = {ib, jb, kb};
						amrex::Box b;
						for (int d = 0; d < 3; ++d) {
							int const len = g.domain.length(d);
							int const base = len / nb[d], rem = len % nb[d];
							b.lo[d] = idx[d] * base + std::min(idx[d], rem);
							b.hi[d] = b.lo[d] + base + (idx[d] < rem ? 1 : 0) - 1;
						}
						grids_.push_back(b);
					}
				}
			}
		}
		// the whole level and its box -> rank map (every rank computes the same); this rank keeps the boxes it owns, in global order
		allGrids_ = grids_;
		if (spec != nullptr) {
			owner_ = spec->owner.empty() ? std::vector<int>(allGrids_.size(), 0) : spec->owner;
			AMREX_ALWAYS_ASSERT(owner_.size() == allGrids_.size());
		} else {
			// an AMR hierarchy keeps every refined box on the rank of its level-0 ancestor: "interleaved" (rank = Morton index of the level-0 box
			// mod nranks; the default there, as in quokka_amd/amr_simulation.py) lands a refined region on every rank, "bricks" keeps level 0
			// compact (fewest remote ghost strips: the uniform-grid default)
			int maxLevel = 0;
			amrex::ParmParse("amr").query("max_level", maxLevel);
			std::string how = (maxLevel > 0 && comm.size > 1) ? "interleaved" : "bricks";
			amrex::ParmParse("qk").query("level0_distribution", how);
			owner_ = (how == "interleaved") ? qkhost::distributeBoxesInterleaved(nb, comm.size) : qkhost::distributeBoxes(nb, comm.size);
		}
		allBoxes_.clear();
		for (auto const &b : allGrids_) {
			allBoxes_.push_back({{b.lo[0], b.lo[1], b.lo[2]}, {b.hi[0], b.hi[1], b.hi[2]}});
		}
		grids_.clear();
		std::vector<qk_box> qb;
		for (size_t n = 0; n < allGrids_.size(); ++n) {
			if (owner_[n] == comm.rank) {
				grids_.push_back(allGrids_[n]);
				qb.push_back(allBoxes_[n]);
			}
		}
		if (grids_.empty() && spec == nullptr) { // (a refined level may well have no box on this rank: every operator on it is then a no-op)
			amrex::Abort("this rank owns no boxes: fewer boxes than ranks (lower amr.max_grid_size)");
		}
		qkhost::check(qk_level_create(rt.ctx, &myLev_, AMREX_SPACEDIM, static_cast<int>(qb.size()), qb.data()), "qk_level_create");
		rt.lev = myLev_;
		int const nc = Physics_Indices<problem_t>::nvarTotal_cc;
		state_new_cc_[0].define(grids_, nc, nghost_cc_);
		state_old_cc_[0].define(grids_, nc, nghost_cc_);
		// ghost-exchange plan: same-rank copies, strips packed for / unpacked from the peers, physical-boundary shells
		for (int d = 0; d < 3; ++d) {
			qgeom_.domain.lo[d] = g.domain.lo[d];
			qgeom_.domain.hi[d] = g.domain.hi[d];
			qgeom_.periodic[d] = g.periodic[d];
		}
		qgeom_.ndim = AMREX_SPACEDIM;
		qkhost::check(qk_ghost_plan_create(myLev_, &plan_, &qgeom_, nghost_cc_, nc, static_cast<int>(allBoxes_.size()), allBoxes_.data(), owner_.data(),
						   comm.rank),
			      "qk_ghost_plan_create");
		peers_.build(plan_, sizeof(double));
	}
	// level description shared by every plan of this level
	std::vector<amrex::Box> allGrids_;
	std::vector<qk_box> allBoxes_;
	std::vector<int> owner_;
	qk_geometry qgeom_{};
	qkhost::PeerBuffers peers_;

	void readParameters() // reference src/simulation.hpp:541-636 (the keys the config decks use)
	{
		amrex::ParmParse pp;
		pp.query("max_timesteps", maxTimesteps_);
		pp.query("cfl", cflNumber_);
		pp.query("stop_time", stopTime_);
		pp.query("plotfile_interval", plotfileInterval_);
		pp.query("checkpoint_interval", checkpointInterval_);
		pp.query("plotfile_prefix", plot_file);
		pp.query("checkpoint_prefix", chk_file);
		pp.query("restartfile", restart_chkfile);
		pp.query("density_floor", densityFloor_);
		pp.query("temperature_floor", tempFloor_);
	}

	[[nodiscard]] auto CountCells(int /*lev*/) const -> amrex::Long // (all ranks)
	{
		amrex::Long n = 0;
		for (auto const &b : allGrids_) {
			n += b.numPts();
		}
		return n;
	}

	// user hooks (specialised per problem)
	virtual void setInitialConditionsOnGrid(quokka::grid const &grid_elem) = 0;
	virtual void preCalculateInitialConditions() {}
	virtual void computeAfterEvolve(amrex::Vector<amrex::Real> & /*initSumCons*/) {}

	// reference src/simulation.hpp:1608-1626
	void setInitialConditions()
	{
		preCalculateInitialConditions();
		auto &mf = state_new_cc_[0];
		if (restart_chkfile.empty()) {
			for (int b = 0; b < mf.size(); ++b) {
				// device mode: the problem's ParallelFor runs as a kernel on the level's own arrays
				quokka::grid grid_elem{mf.array(b), mf.validbox(b), geom[0].CellSizeArray(), geom[0].ProbLoArray(), geom[0].ProbHiArray()};
				setInitialConditionsOnGrid(grid_elem);
			}
		} else {
			// level 0 of ReadCheckpointFile (reference src/simulation.hpp:2736-2801): the BoxArray comes from the deck, the data by
			// ParallelCopy from the file's boxes
			auto const h = quokka::io::ReadCheckpointHeader(restart_chkfile);
			istep[0] = h.istep.at(0);
			dt_[0] = h.dt.at(0);
			tNew_[0] = h.tNew.at(0);
			quokka::io::VisMFReadInto(mf, restart_chkfile + "/Level_0/Cell");
		}
		fillBoundaryConditions(state_new_cc_[0]);
		amrex::MultiFab::Copy(state_old_cc_[0], state_new_cc_[0]);
		if (restart_chkfile.empty()) {
			setInitialConditionsAtLevel_fc();
		} else {
			readFaceCentredState();
		}
		areInitialConditionsDefined_ = true;
	}
	// the face-centred part of ReadCheckpointFile (reference src/simulation.hpp:2779-2815)
	void readFaceCentredState()
	{
		if constexpr (qkhost::hasFaceState<problem_t>()) {
			defineFaceCentredState();
			char const *dirName[3] = {"x", "y", "z"};
			for (int idim = 0; idim < AMREX_SPACEDIM; ++idim) {
				quokka::io::VisMFReadInto(state_new_fc_[0][idim], restart_chkfile + "/Level_0/Face_" + dirName[idim]);
				amrex::MultiFab::Copy(state_old_fc_[0][idim], state_new_fc_[0][idim]);
			}
		}
	}
	// setInitialConditionsAtLevel_fc (reference src/simulation.hpp:1628-1651): the face-centred state of problems that carry one
	// (Physics_Indices::nvarTotal_fc > 0: face velocities, the magnetic field of the MHD index bookkeeping).  The arrays exist, take the
	// problem's initial conditions and travel through checkpoints and plotfiles; their ghost faces are NOT filled — no operator of this host
	// reads them (the reference's MHD update does not exist either: hydro/mhd_system.hpp holds indices only).
	void defineFaceCentredState()
	{
		if constexpr (qkhost::hasFaceState<problem_t>()) {
			if (state_new_fc_.empty()) {
				state_new_fc_.resize(1);
				state_old_fc_.resize(1);
				for (int idim = 0; idim < AMREX_SPACEDIM; ++idim) {
					state_new_fc_[0][idim].define(grids_, Physics_Indices<problem_t>::nvarPerDim_fc, nghost_fc_, idim);
					state_old_fc_[0][idim].define(grids_, Physics_Indices<problem_t>::nvarPerDim_fc, nghost_fc_, idim);
				}
			}
		}
	}
	void setInitialConditionsAtLevel_fc()
	{
		if constexpr (qkhost::hasFaceState<problem_t>()) {
			defineFaceCentredState();
			for (int idim = 0; idim < AMREX_SPACEDIM; ++idim) {
				auto &mf = state_new_fc_[0][idim];
				mf.setVal(0.);
				for (int b = 0; b < mf.size(); ++b) {
					amrex::Box faces = mf.validbox(b); // iter.validbox() of a face-centred MultiFab: nodal in idim
					faces.hi[idim] += 1;
					quokka::grid grid_elem{mf.array(b),	  faces, geom[0].CellSizeArray(), geom[0].ProbLoArray(), geom[0].ProbHiArray(), quokka::centering::fc,
							       static_cast<quokka::direction>(idim)};
					setInitialConditionsOnGridFaceVars(grid_elem);
				}
				amrex::MultiFab::Copy(state_old_fc_[0][idim], mf);
			}
		}
	}
	virtual void setInitialConditionsOnGridFaceVars(quokka::grid const & /*grid_elem*/) {}
	// componentNames_fc_ (reference src/QuokkaSimulation.hpp:310-321: the face velocities of every direction, then the field components — the
	// order of the reference's labels, kept although PlotFileMFAtLevel stores the averages direction by direction)
	[[nodiscard]] static auto componentNames_fc() -> std::vector<std::string>
	{
		char const *dirName[3] = {"x", "y", "z"};
		std::vector<std::string> names;
		if constexpr (qkhost::hasFaceState<problem_t>()) {
			if constexpr (Physics_Traits<problem_t>::is_hydro_enabled) {
				for (int idim = 0; idim < AMREX_SPACEDIM; ++idim) {
					names.push_back(std::string(dirName[idim]) + "-velocity");
				}
			}
			if constexpr (Physics_Traits<problem_t>::is_mhd_enabled) {
				for (int idim = 0; idim < AMREX_SPACEDIM; ++idim) {
					names.push_back(std::string(dirName[idim]) + "-BField");
				}
			}
		}
		return names;
	}
	[[nodiscard]] auto getNewMF_fc() const -> amrex::Vector<amrex::Array<amrex::MultiFab, AMREX_SPACEDIM>> const & { return state_new_fc_; }
	void setChkFile(std::string const &chkfile_number) { restart_chkfile = chkfile_number; } // reference src/simulation.hpp:410
	amrex::Vector<amrex::Array<amrex::MultiFab, AMREX_SPACEDIM>> state_new_fc_, state_old_fc_;
	amrex::Vector<amrex::BCRec> BCs_fc_;
	int nghost_fc_ = Physics_Traits<problem_t>::is_mhd_enabled ? 4 : 2; // reference src/simulation.hpp:364

	// setInitialConditionsAtLevel_cc (reference src/simulation.hpp:1608-1626): the problem's initial conditions on this level's boxes
	void setInitialConditionsAtLevel()
	{
		auto &mf = state_new_cc_[0];
		for (int b = 0; b < mf.size(); ++b) {
			quokka::grid grid_elem{mf.array(b), mf.validbox(b), geom[0].CellSizeArray(), geom[0].ProbLoArray(), geom[0].ProbHiArray()};
			setInitialConditionsOnGrid(grid_elem);
		}
		amrex::MultiFab::Copy(state_old_cc_[0], state_new_cc_[0]);
		setInitialConditionsAtLevel_fc();
		areInitialConditionsDefined_ = true;
		newStateGhostsFilled_ = false;
	}

	// fillBoundaryConditions for the radiation transport kernels, which read only the radiation components of the ghost cells
	void fillRadiationGhosts(amrex::MultiFab &state)
	{
		int const first = Physics_Indices<problem_t>::radFirstIndex;
		qkhost::check(qk_ghost_plan_set_components(plan_, first, state.nComp() - first), "qk_ghost_plan_set_components");
		fillBoundaryConditions(state);
		qkhost::check(qk_ghost_plan_set_components(plan_, 0, -1), "qk_ghost_plan_set_components");
	}
	// BCs_cc_ as the C-ABI takes them
	[[nodiscard]] auto boundaryRecords() const -> std::vector<qk_bcrec>
	{
		std::vector<qk_bcrec> bcs(BCs_cc_.size());
		for (size_t n = 0; n < BCs_cc_.size(); ++n) {
			for (int d = 0; d < 3; ++d) {
				bcs[n].lo[d] = (d < AMREX_SPACEDIM) ? BCs_cc_[n].lo(d) : 0;
				bcs[n].hi[d] = (d < AMREX_SPACEDIM) ? BCs_cc_[n].hi(d) : 0;
			}
		}
		return bcs;
	}
	// level-0 branch of fillBoundaryConditions (reference src/simulation.hpp:1751-1776).
	// `between` (optional; the multi-GPU schedule of north_star): called while the strips of the other ranks are on the wire — RCCL moves them on
	// its own HIP stream (qk_comm.hpp) —, after the boxes that receive nothing remote have been completed (same-rank copies + their own
	// physical-boundary slabs).  The caller advances exactly those boxes in it; the rest follows after the unpack.
	void fillBoundaryConditions(amrex::MultiFab &state, std::function<void()> const &between = {})
	{
		activate();
		hipStream_t const cs = qkhost::Runtime::get().computeStream();
		// state.FillBoundary(geom.periodicity()) (reference src/simulation.hpp:1755): strips for the other ranks are packed, sent peer to peer
		// while the same-rank copies run, and unpacked
		for (size_t k = 0; k < peers_.peer.size(); ++k) {
			qkhost::check(qk_FillBoundary_pack(plan_, cs, static_cast<int>(k), qkhost::tab(state), static_cast<double *>(peers_.send[k])),
				      "FillBoundary_pack");
		}
		qkhost::Comm::get().exchangeBegin(peers_.peer, peers_.send, peers_.nsend, peers_.recv, peers_.nrecv, sizeof(double), cs);
		qkhost::check(qk_FillBoundary_local(plan_, cs, qkhost::tab(state)), "FillBoundary");
		bool const physical = !geom[0].isAllPeriodic();
		auto const bcs = boundaryRecords();
		auto physbc = [&](int which) {
			if (physical) {
				qkhost::check(qk_FillPhysicalBoundary_subset(plan_, cs, qkhost::tab(state), bcs.data(), nullptr, which), "FillPhysicalBoundary");
				customBoundaryConditionsOnDevice(state, which);
			}
		};
		if (between) {
			AMREX_ALWAYS_ASSERT(!beforePhysBC_); // (a refined level interpolates its uncovered ghost cells first: no split there)
			physbc(QK_BOXES_LOCAL_ONLY);
			between();
		}
		qkhost::Comm::get().exchangeEnd(cs);
		for (size_t k = 0; k < peers_.peer.size(); ++k) {
			qkhost::check(qk_FillBoundary_unpack(plan_, cs, static_cast<int>(k), qkhost::tab(state), static_cast<const double *>(peers_.recv[k])),
				      "FillBoundary_unpack");
		}
		if (beforePhysBC_) {
			beforePhysBC_(state);
		}
		physbc(between ? QK_BOXES_REMOTE_DEPENDENT : QK_BOXES_ALL);
	}
	// The physical-boundary rules alone (PhysBCFunct + the problem's hook) on one group of boxes of `state` — QK_BOXES_LOCAL_ONLY: the boxes not marked
	// by setBoxGroups —: what the children of a speculative coarse step read of the parent's new state beyond its near boxes lies beyond the domain.
	void fillPhysicalBoundaries(amrex::MultiFab &state, int which)
	{
		activate();
		if (geom[0].isAllPeriodic()) {
			return;
		}
		auto const bcs = boundaryRecords();
		qkhost::check(qk_FillPhysicalBoundary_subset(plan_, qkhost::Runtime::get().computeStream(), qkhost::tab(state), bcs.data(), nullptr, which), "FillPhysicalBoundary");
		customBoundaryConditionsOnDevice(state, which);
	}
	// marks the boxes of the second group ("remote" in the ghost plan's vocabulary: their physical-boundary slabs are filled by
	// QK_BOXES_REMOTE_DEPENDENT, the others' by QK_BOXES_LOCAL_ONLY); the slab lists cached per group are dropped
	void setBoxGroups(std::vector<char> const &second)
	{
		for (int b = 0; b < static_cast<int>(second.size()); ++b) {
			qkhost::check(qk_ghost_plan_set_box_remote(plan_, b, second[b] != 0 ? 1 : 0), "qk_ghost_plan_set_box_remote");
		}
		for (auto it = bcShells_.begin(); it != bcShells_.end();) {
			if (it->first.second != QK_BOXES_ALL) {
				(void)hipFree(it->second.d);
				it = bcShells_.erase(it);
			} else {
				++it;
			}
		}
	}
	// setCustomBoundaryConditions as the reference runs it (simulation.hpp:297-299, :1550-1561; amrex::GpuBndryFuncFab): the problem's
	// DEVICE function is called for every ghost cell that lies outside the domain in a non-periodic direction, after the mathematical
	// boundary types have been filled.  One kernel instantiated with the problem type per box — arbitrary boundary code, not the closed
	// Dirichlet / Marshak set of the C-ABI (which host-mode problems are sampled into).
	void customBoundaryConditionsOnDevice(amrex::MultiFab &state, int which = QK_BOXES_ALL)
	{
		customBoundaryConditionsOn(state, which, plan_, geom[0], bcShells_, bcFillTime());
	}
	// ... on any array of cells of a level with geometry `g` (this level's own state, or the coarse patch a child on another rank keeps of it:
	// AmrDriver::Shadow, whose slab lists live in `cache`)
	void customBoundaryConditionsOn(amrex::MultiFab &state, int which, qk_ghost_plan *plan, amrex::Geometry const &g, BcShellCache &cache, double time)
	{
		if (customBcIsDefault_ == 1) {
			return; // the hook is the default (an empty body): known since the first fill
		}
		if (d_bcrec_ == nullptr) {
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_bcrec_), sizeof(amrex::BCRec) * BCs_cc_.size()));
			QK_HOST_HIP(hipMemcpy(d_bcrec_, BCs_cc_.data(), sizeof(amrex::BCRec) * BCs_cc_.size(), hipMemcpyHostToDevice));
		}
		auto const gd = g.data();
		// The slabs of ghost cells beyond the non-periodic faces, disjoint (x slabs over the whole y-z extent of the fab, y slabs over the x range
		// inside the domain, z slabs over the x and y ranges inside), listed once per ghost width and box group: every cell-centred array of a
		// level has the same boxes.  It was one whole-fab launch per box and fill (62 per coarse step of the config-5 hierarchy, each testing
		// 2.5 M cells to find 0.3 M).
		auto const key = std::make_pair(state.nGrow(), which);
		auto it = cache.find(key);
		if (it == cache.end()) {
			std::vector<qkhost::BcShell> h;
			amrex::Long most = 0;
			for (int b = 0; b < state.size(); ++b) {
				if (which != QK_BOXES_ALL && (qk_ghost_plan_box_is_remote(plan, b) == 1) != (which == QK_BOXES_REMOTE_DEPENDENT)) {
					continue; // the other group of an overlapped fill
				}
				amrex::Box rest = state.fabbox(b);
				for (int d = 0; d < AMREX_SPACEDIM; ++d) {
					if (g.isPeriodic(d)) {
						continue;
					}
					for (int side = 0; side < 2; ++side) {
						amrex::Box sl = rest;
						if (side == 0) {
							sl.hi[d] = std::min(rest.hi[d], gd.domain.lo[d] - 1);
						} else {
							sl.lo[d] = std::max(rest.lo[d], gd.domain.hi[d] + 1);
						}
						if (sl.ok()) {
							qkhost::BcShell e{b, {sl.lo[0], sl.lo[1], sl.lo[2]}, {sl.hi[0], sl.hi[1], sl.hi[2]}};
							h.push_back(e);
							most = std::max(most, sl.numPts());
						}
					}
					rest.lo[d] = std::max(rest.lo[d], gd.domain.lo[d]);
					rest.hi[d] = std::min(rest.hi[d], gd.domain.hi[d]);
				}
			}
			BcShellList l;
			l.count = static_cast<int>(h.size());
			l.most = most;
			if (l.count > 0) {
				QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&l.d), sizeof(qkhost::BcShell) * h.size()));
				QK_HOST_HIP(hipMemcpy(l.d, h.data(), sizeof(qkhost::BcShell) * h.size(), hipMemcpyHostToDevice));
			}
			it = cache.emplace(key, l).first;
		}
		auto const &l = it->second;
		if (l.count == 0) {
			return;
		}
		if (customBcIsDefault_ < 0) { // (the flag is one per executable: another problem type's object may have raised it)
			int const zero = 0;
			QK_HOST_HIP(hipStreamSynchronize(qkhost::Runtime::get().computeStream()));
			QK_HOST_HIP(hipMemcpyToSymbol(HIP_SYMBOL(qkhost::g_defaultCustomBcRan), &zero, sizeof(int)));
		}
		hipLaunchKernelGGL(qkhost::customBcKernel<problem_t>, dim3(static_cast<unsigned>((l.most + 255) / 256), static_cast<unsigned>(l.count)), dim3(256), 0,
				   qkhost::Runtime::get().computeStream(), l.d, state.arrays(), gd, time, d_bcrec_, state.nComp());
		if (customBcIsDefault_ < 0) { // first launch: did the default body run?
			int ran = 0;
			QK_HOST_HIP(hipStreamSynchronize(qkhost::Runtime::get().computeStream()));
			QK_HOST_HIP(hipMemcpyFromSymbol(&ran, HIP_SYMBOL(qkhost::g_defaultCustomBcRan), sizeof(int)));
			customBcIsDefault_ = (ran != 0) ? 1 : 0;
		}
	}
	int customBcIsDefault_ = -1; // -1 unknown, 0 the problem specialised setCustomBoundaryConditions, 1 it is the empty default
	BcShellCache bcShells_; // (ghost width, box group) -> slabs; lives as long as the level object
	amrex::BCRec *d_bcrec_ = nullptr;
	[[nodiscard]] virtual auto bcFillTime() const -> double { return tNew_[0]; }
	// set by a level's advance when the ghost cells of state_old_cc_ hold the fill at the old time already (AmrDriver then skips its own fill of
	// the old state before the children interpolate from it)
	bool oldStateGhostsFilled_ = false;
	// the ghost cells of state_new_cc_ hold the fill at the level's new time (set by AmrDriver::fillGhosts, cleared by whatever writes the state).
	// Every clearing also counts a new VERSION of the level's states: the copies of them that the children of a hierarchy with distributed levels
	// keep on their own ranks (AmrDriver::Shadow) are fetched once per version.
	struct GhostFlag {
		bool filled = false;
		std::uint64_t version = 0;
		auto operator=(bool v) -> GhostFlag &
		{
			filled = v;
			if (!v) {
				++version;
			}
			return *this;
		}
		operator bool() const { return filled; } // NOLINT
	};
	GhostFlag newStateGhostsFilled_;

      protected:
	qk_level *myLev_ = nullptr;
	qk_ghost_plan *plan_ = nullptr;
};


#endif // QK_HOST_QUOKKA_AMR_SIMULATION_HPP_
