// general_opacity.cpp — builder-authored problem for the compiled hook path of the C++ host (no reference problem is its origin): a periodic
// 1-D radiating flow whose opacity law
//         kappa_P = kappa_E = kappa_F = kappa0 rho^0.3 (T / T0)^-1.7
// lies outside every closed opacity set of the C-ABI.  The three opacity hooks below are compiled into the source-term kernel of THIS
// translation unit (quokka_amd/host/qk_problem_kernels.hpp) and evaluated inside its Newton-Raphson iteration; the CPU oracle runs the same
// problem with the same expression as a std::function (oracle/problems.hpp setupGeneralOpacity, tests/test_compiled_hooks_gpu.py).
#include "QuokkaSimulation.hpp"
#include "radiation/radiation_system.hpp"

struct GeneralOpacity {
};

namespace
{
constexpr double c = 1.0e8, chat = 1.0e7, a_rad = 1.0, mu = 1.0, k_B = 1.0;
constexpr double rho0 = 1.0, T0 = 1.0, kappa0 = 2.0e-3, L = 64.0, v0 = 1.0e-3 * c;
} // namespace

template <> struct quokka::EOS_Traits<GeneralOpacity> {
	static constexpr double mean_molecular_weight = mu;
	static constexpr double boltzmann_constant = k_B;
	static constexpr double gamma = 5. / 3.;
};
template <> struct Physics_Traits<GeneralOpacity> {
	static constexpr bool is_hydro_enabled = true;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = 0;
	static constexpr bool is_radiation_enabled = true;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};
template <> struct RadSystem_Traits<GeneralOpacity> {
	static constexpr double c_light = c;
	static constexpr double c_hat = chat;
	static constexpr double radiation_constant = a_rad;
	static constexpr double Erad_floor = 0.;
	static constexpr int beta_order = 1;
};

AMREX_GPU_HOST_DEVICE inline auto opacityLaw(double rho, double T) -> double { return kappa0 * std::pow(rho, 0.3) * std::pow(T / T0, -1.7); }
template <> AMREX_GPU_HOST_DEVICE auto RadSystem<GeneralOpacity>::ComputePlanckOpacity(double rho, double Tgas) -> amrex::Real { return opacityLaw(rho, Tgas); }
template <> AMREX_GPU_HOST_DEVICE auto RadSystem<GeneralOpacity>::ComputeFluxMeanOpacity(double rho, double Tgas) -> amrex::Real { return opacityLaw(rho, Tgas); }
template <> AMREX_GPU_HOST_DEVICE auto RadSystem<GeneralOpacity>::ComputeEnergyMeanOpacity(double rho, double Tgas) -> amrex::Real { return opacityLaw(rho, Tgas); }

template <> void QuokkaSimulation<GeneralOpacity>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	auto const dx = grid_elem.dx_;
	auto const prob_lo = grid_elem.prob_lo_;
	auto const &state_cc = grid_elem.array_;
	amrex::ParallelFor(grid_elem.indexRange_, [=] AMREX_GPU_DEVICE(int i, int j, int k) {
		double const twopi = 2.0 * 3.14159265358979323846;
		double const x = prob_lo[0] + (i + 0.5) * dx[0];
		double const s = std::sin(twopi * x / L), co = std::cos(twopi * x / L);
		double const rho = rho0 * (1.0 + 0.3 * s);
		double const T = T0 * (1.0 + 0.2 * co);
		double const Trad = T0 * (1.0 - 0.1 * s);
		double const Egas = quokka::EOS<GeneralOpacity>::ComputeEintFromTgas(rho, T);
		double const erad = a_rad * ((Trad * Trad) * (Trad * Trad));
		state_cc(i, j, k, RadSystem<GeneralOpacity>::radEnergy_index) = erad;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::x1RadFlux_index) = 0.05 * c * erad * co;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::x2RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::x3RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::gasEnergy_index) = Egas + 0.5 * rho * v0 * v0;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::gasDensity_index) = rho;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::gasInternalEnergy_index) = Egas;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::x1GasMomentum_index) = v0 * rho;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::x2GasMomentum_index) = 0.;
		state_cc(i, j, k, RadSystem<GeneralOpacity>::x3GasMomentum_index) = 0.;
	});
}

auto problem_main() -> int
{
	constexpr int nvars = RadSystem<GeneralOpacity>::nvar_;
	amrex::Vector<amrex::BCRec> BCs_cc(nvars);
	for (int n = 0; n < nvars; ++n) {
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			BCs_cc[n].setLo(d, amrex::BCType::int_dir); // periodic
			BCs_cc[n].setHi(d, amrex::BCType::int_dir);
		}
	}
	QuokkaSimulation<GeneralOpacity> sim(BCs_cc);
	sim.reconstructionOrder_ = 3;
	sim.radiationReconstructionOrder_ = 3;
	sim.stopTime_ = 1.0e300;
	sim.radiationCflNumber_ = 0.3;
	sim.cflNumber_ = 0.3;
	sim.maxTimesteps_ = 40;
	sim.setInitialConditions();
	sim.evolve();
	return 0;
}
