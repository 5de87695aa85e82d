// sedov_bench.cpp — builder-authored timing driver of the C++17 host (bench.py's `cxx_host` block): the BASELINE metric — cell-updates per second
// of the 3-D Sedov blast on a uniform grid (deck decks/blast_unigrid_256.in: 256^3 in 128^3 boxes, gamma = 1.4, PPM + HLLC RK2, CFL 0.3,
// reflecting octant) — measured through QuokkaSimulation<problem_t> exactly as a problem file drives it.  W warm-up steps (a first evolve()), then
// K timed steps (a second evolve(): its step loop between two device synchronisations); one JSON line on rank 0.  One process per GPU under any launcher that exports
// RANK / WORLD_SIZE / LOCAL_RANK (qk_comm.hpp).  Deck keys of this driver: bench.warmup, bench.steps.
#include <chrono>

#include "QuokkaSimulation.hpp"
#include "hydro/hydro_system.hpp"

struct BlastBench {
};

template <> struct quokka::EOS_Traits<BlastBench> {
	static constexpr double gamma = 1.4;
	static constexpr double mean_molecular_weight = C::m_u;
	static constexpr double boltzmann_constant = C::k_B;
};
template <> struct HydroSystem_Traits<BlastBench> {
	static constexpr bool reconstruct_eint = false;
};
template <> struct Physics_Traits<BlastBench> {
	static constexpr bool is_hydro_enabled = true;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = 0;
	static constexpr bool is_radiation_enabled = false;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

template <> void QuokkaSimulation<BlastBench>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	// the whole blast energy in the corner cell of the octant, 1e-10 of that energy density elsewhere; gas at rest, unit density
	auto const dx = grid_elem.dx_;
	auto const &state = grid_elem.array_;
	double const cell_vol = dx[0] * dx[1] * dx[2];
	double const rho_e_blast = (0.851072 / 8.0) / cell_vol;
	amrex::ParallelFor(grid_elem.indexRange_, [=] AMREX_GPU_DEVICE(int i, int j, int k) {
		double const rho_e = (i == 0 && j == 0 && k == 0) ? rho_e_blast : 1.0e-10 * rho_e_blast;
		for (int n = 0; n < state.nComp(); ++n) {
			state(i, j, k, n) = 0.;
		}
		state(i, j, k, HydroSystem<BlastBench>::density_index) = 1.0;
		state(i, j, k, HydroSystem<BlastBench>::energy_index) = rho_e;
	});
}

auto problem_main() -> int
{
	constexpr int ncomp = Physics_Indices<BlastBench>::nvarTotal_cc;
	amrex::Vector<amrex::BCRec> BCs_cc(ncomp);
	for (int n = 0; n < ncomp; ++n) {
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			bool const normal_momentum = (n == HydroSystem<BlastBench>::x1Momentum_index + d);
			BCs_cc[n].setLo(d, normal_momentum ? amrex::BCType::reflect_odd : amrex::BCType::reflect_even);
			BCs_cc[n].setHi(d, normal_momentum ? amrex::BCType::reflect_odd : amrex::BCType::reflect_even);
		}
	}
	int warmup = 3, steps = 20;
	amrex::ParmParse pb("bench");
	pb.query("warmup", warmup);
	pb.query("steps", steps);

	QuokkaSimulation<BlastBench> sim(BCs_cc);
	sim.reconstructionOrder_ = 3;
	sim.stopTime_ = 1.0e300;
	sim.cflNumber_ = 0.3;
	sim.plotfileInterval_ = -1;
	sim.setInitialConditions();

	auto &comm = qkhost::Comm::get();
	sim.maxTimesteps_ = warmup;
	sim.evolve();
	QK_HOST_HIP(hipDeviceSynchronize());
	comm.barrier();
	sim.maxTimesteps_ = warmup + steps;
	sim.evolve();
	// evolve() times its step loop between two device synchronisations (elapsedSeconds_: the reference's figure of merit, simulation.hpp:972-977;
	// its conservation sums and reports before and after the loop are outside); the slowest rank counts
	double elapsed = comm.allReduceMax(sim.elapsedSeconds_);
	double const cells = static_cast<double>(sim.CountCells(0));
	if (comm.rank == 0) {
		std::printf("{\"cxx_host\": true, \"value\": %.6f, \"unit\": \"Mcell-updates/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6f, "
			    "\"cells\": %.0f, \"fofc_stages\": %ld, \"retries\": %ld, \"sim_time\": %.17g}\n",
			    cells * steps / elapsed / 1.0e6, comm.size, steps, warmup, elapsed / steps * 1.0e3, cells, sim.fofcStages_, sim.retries_, sim.tNew_[0]);
	}
	return 0;
}
