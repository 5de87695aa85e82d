// Marshak wave in the equilibrium-diffusion limit (McClarren & Lowrie 2008) in 1-D — problem generator written against the reference's
// surface (cf. reference src/problems/RadMarshakAsymptotic/test_radiation_marshak_asymptotic.cpp; deck tests/MarshakAsymptotic.in).
// Radiation only, gamma-law gas, absorption coefficient 300 (T / T_H)^-3 per cm, Eddington approximation, Marshak half-range condition
// on the lower face.  Exit status = the reference's pass criterion: relative L1 error of the gas temperature against the similarity
// solution (extern/marshak_similarity.csv, path from the deck: `marshak.solution_file`) below 9 per cent.
#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "AMReX_BC_TYPES.H"
#include "AMReX_ParmParse.H"
#include "AMReX_Print.H"

#include "QuokkaSimulation.hpp"
#include "hydro/EOS.hpp"
#include "math/interpolate.hpp"
#include "radiation/radiation_system.hpp"

struct SuOlsonProblemCgs {
};

constexpr double kappa = 300.0;		      // cm^-1 (absorption coefficient at T_H)
constexpr double rho0 = 2.0879373766122384;   // g cm^-3
constexpr double T_hohlraum = 1.1604448449e7; // K (1 keV)
constexpr double T_initial = T_hohlraum * 0.001;
constexpr double a_rad = radiation_constant_cgs_;
constexpr double Erad_floor_ = a_rad * T_initial * T_initial * T_initial * T_initial;

template <> struct quokka::EOS_Traits<SuOlsonProblemCgs> {
	static constexpr double mean_molecular_weight = C::m_u;
	static constexpr double boltzmann_constant = C::k_B;
	static constexpr double gamma = 5. / 3.;
	static constexpr double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
};

template <> struct RadSystem_Traits<SuOlsonProblemCgs> {
	static constexpr double c_light = c_light_cgs_;
	static constexpr double c_hat = c_light_cgs_;
	static constexpr double radiation_constant = radiation_constant_cgs_;
	static constexpr double Erad_floor = Erad_floor_;
	static constexpr int beta_order = 0;
};

template <> struct Physics_Traits<SuOlsonProblemCgs> {
	static constexpr bool is_hydro_enabled = false;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 0;
	static constexpr bool is_radiation_enabled = true;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

template <> auto RadSystem<SuOlsonProblemCgs>::ComputePlanckOpacity(const double rho, const double Tgas) -> amrex::Real
{
	auto sigma = kappa * std::pow(Tgas / T_hohlraum, -3); // cm^-1
	return sigma / rho;				      // cm^2 g^-1
}
template <> auto RadSystem<SuOlsonProblemCgs>::ComputeFluxMeanOpacity(const double rho, const double Tgas) -> amrex::Real
{
	return ComputePlanckOpacity(rho, Tgas);
}
template <> auto RadSystem<SuOlsonProblemCgs>::ComputeEddingtonFactor(double /*f*/) -> double
{
	return (1. / 3.); // Eddington approximation
}

template <>
void AMRSimulation<SuOlsonProblemCgs>::setCustomBoundaryConditions(const amrex::IntVect &iv, amrex::Array4<amrex::Real> const &consVar, int /*dcomp*/,
								   int /*numcomp*/, amrex::GeometryData const & /*geom*/, const amrex::Real /*time*/,
								   const amrex::BCRec * /*bcr*/, int /*bcomp*/, int /*orig_comp*/)
{
	auto const i = iv.toArray()[0];
	int const j = 0, k = 0;
	if (i < 0) {
		// Marshak condition (first-order accurate: the ghost flux follows the first cell inside the face)
		const double E_inc = radiation_constant_cgs_ * std::pow(T_hohlraum, 4);
		const double c = c_light_cgs_;
		const double E_0 = consVar(0, j, k, RadSystem<SuOlsonProblemCgs>::radEnergy_index);
		const double F_0 = consVar(0, j, k, RadSystem<SuOlsonProblemCgs>::x1RadFlux_index);
		const double F_bdry = 0.5 * c * E_inc - 0.5 * (c * E_0 + 2.0 * F_0);
		consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::radEnergy_index) = E_inc;
		consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x1RadFlux_index) = F_bdry;
		consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x2RadFlux_index) = 0.;
		consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x3RadFlux_index) = 0.;
	} else {
		consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::radEnergy_index) = radiation_constant_cgs_ * std::pow(T_initial, 4);
		consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x1RadFlux_index) = 0;
		consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x2RadFlux_index) = 0;
		consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x3RadFlux_index) = 0;
	}
	const double Egas = quokka::EOS<SuOlsonProblemCgs>::ComputeEintFromTgas(rho0, T_initial);
	consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::gasEnergy_index) = Egas;
	consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::gasDensity_index) = rho0;
	consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::gasInternalEnergy_index) = Egas;
	consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x1GasMomentum_index) = 0.;
	consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x2GasMomentum_index) = 0.;
	consVar(i, j, k, RadSystem<SuOlsonProblemCgs>::x3GasMomentum_index) = 0.;
}

template <> void QuokkaSimulation<SuOlsonProblemCgs>::setInitialConditionsOnGrid(quokka::grid const &grid_elem)
{
	const amrex::Array4<double> &state_cc = grid_elem.array_;
	amrex::ParallelFor(grid_elem.indexRange_, [=](int i, int j, int k) {
		const double Egas = quokka::EOS<SuOlsonProblemCgs>::ComputeEintFromTgas(rho0, T_initial);
		const double Erad = a_rad * std::pow(T_initial, 4);
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::radEnergy_index) = Erad;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::x1RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::x2RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::x3RadFlux_index) = 0;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::gasDensity_index) = rho0;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::gasEnergy_index) = Egas;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::gasInternalEnergy_index) = Egas;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::x1GasMomentum_index) = 0.;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::x2GasMomentum_index) = 0.;
		state_cc(i, j, k, RadSystem<SuOlsonProblemCgs>::x3GasMomentum_index) = 0.;
	});
}

auto problem_main() -> int
{
	const int max_timesteps = 1e6;
	const double CFL_number = 10.0;
	const double initial_dt = 5.0e-12; // s
	const double max_dt = 5.0;	   // s
	const double max_time = 10.0e-9;   // s

	constexpr int nvars = RadSystem<SuOlsonProblemCgs>::nvar_;
	amrex::Vector<amrex::BCRec> BCs_cc(nvars);
	for (int n = 0; n < nvars; ++n) {
		BCs_cc[n].setLo(0, amrex::BCType::ext_dir);  // custom (Marshak) x1
		BCs_cc[n].setHi(0, amrex::BCType::foextrap); // extrapolate x1
	}

	QuokkaSimulation<SuOlsonProblemCgs> sim(BCs_cc);
	sim.radiationReconstructionOrder_ = 3; // PPM
	sim.stopTime_ = max_time;
	sim.initDt_ = initial_dt;
	sim.maxDt_ = max_dt;
	sim.radiationCflNumber_ = CFL_number;
	sim.maxTimesteps_ = max_timesteps;
	sim.plotfileInterval_ = -1;

	sim.setInitialConditions();
	sim.evolve();

	auto const &mf = sim.state_new_cc_[0];
	int const nx = sim.geom[0].Domain().length(0);
	std::vector<double> xs(nx), Tgas_keV(nx);
	for (int b = 0; b < mf.size(); ++b) {
		auto h = mf.copyToHost(b);
		amrex::Array4<double> a(h.data(), mf.fabbox(b), mf.nComp());
		amrex::ParallelFor(mf.validbox(b), [&](int i, int j, int k) {
			xs.at(i) = sim.geom[0].ProbLo(0) + (i + 0.5) * sim.geom[0].CellSize(0);
			const double rho = a(i, j, k, RadSystem<SuOlsonProblemCgs>::gasDensity_index);
			const double x1GasMom = a(i, j, k, RadSystem<SuOlsonProblemCgs>::x1GasMomentum_index);
			const double Ekin = (x1GasMom * x1GasMom) / (2.0 * rho);
			const double Egas_t = a(i, j, k, RadSystem<SuOlsonProblemCgs>::gasEnergy_index) - Ekin;
			Tgas_keV.at(i) = quokka::EOS<SuOlsonProblemCgs>::ComputeTgasFromEint(rho, Egas_t) / T_hohlraum;
		});
	}

	std::string filename = "../extern/marshak_similarity.csv";
	amrex::ParmParse pp("marshak");
	pp.query("solution_file", filename);
	std::ifstream fstream(filename, std::ios::in);
	AMREX_ALWAYS_ASSERT(fstream.is_open());
	std::vector<double> xs_exact, Tmat_exact;
	std::string header;
	std::getline(fstream, header); // (the file has no header line: its first point is not compared, as in the reference)
	for (std::string line; std::getline(fstream, line);) {
		std::istringstream iss(line);
		std::vector<double> row;
		for (double value = NAN; iss >> value;) {
			row.push_back(value);
		}
		if (row.size() < 2) {
			continue;
		}
		xs_exact.push_back(row.at(0));
		Tmat_exact.push_back(row.at(1));
	}

	// numerical solution interpolated onto the tabulated points
	std::vector<double> Tmat_interp(xs_exact.size());
	interpolate_arrays(xs_exact.data(), Tmat_interp.data(), static_cast<int>(xs_exact.size()), xs.data(), Tgas_keV.data(), static_cast<int>(xs.size()));
	double err_norm = 0., sol_norm = 0.;
	for (size_t i = 0; i < xs_exact.size(); ++i) {
		err_norm += std::abs(Tmat_interp[i] - Tmat_exact[i]);
		sol_norm += std::abs(Tmat_exact[i]);
	}
	const double error_tol = 0.09;
	const double rel_error = err_norm / sol_norm;
	sim.errorNorm_ = rel_error;
	amrex::Print() << "Relative L1 error norm = " << rel_error << std::endl;
	qkDumpState(sim);
	return ((rel_error > error_tol) || std::isnan(rel_error)) ? 1 : 0;
}
