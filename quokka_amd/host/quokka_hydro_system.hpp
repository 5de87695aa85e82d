// quokka_hydro_system.hpp — part 1 of the C++17 host mirror (quokka_host.hpp includes the parts in order):
//   Microphysics constants, Physics_Traits / Physics_Indices (reference src/physics_info.hpp:8-47), quokka::EOS_Traits / quokka::EOS (src/hydro/EOS.hpp),
//   HydroSystem_Traits, the runtime singleton (context, level handle, streams), HyperbolicSystem<problem_t> and HydroSystem<problem_t>: static methods with
//   the reference's names and arguments, every body ONE call into the C-ABI (include/quokka_amd.h) — no arithmetic on the host.
#ifndef QK_HOST_QUOKKA_HYDRO_SYSTEM_HPP_
#define QK_HOST_QUOKKA_HYDRO_SYSTEM_HPP_

#include <chrono>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>

#include "amrex_mini.hpp"
#include "compat/planck_integral.hpp"
#include "compat/util_compat.hpp"
#include "qk_comm.hpp"
#include "quokka_io.hpp"

// Microphysics fundamental_constants.H (CODATA 2018, cgs)
namespace C
{
constexpr double k_B = 1.380649e-16;
constexpr double m_u = 1.6605390666e-24;
constexpr double m_p = 1.67262192369e-24;
constexpr double m_e = 9.1093837015e-28;
constexpr double m_n = 1.67492749804e-24;
constexpr double c_light = 2.99792458e10;
constexpr double sigma_SB = 5.670374419e-5;
constexpr double a_rad = 4.0 * sigma_SB / c_light;
constexpr double hplanck = 6.62607015e-27;
constexpr double hbar = 1.054571817e-27;
constexpr double n_A = 6.02214076e23;
constexpr double q_e = 4.80320471e-10;
constexpr double Gconst = 6.67430e-8;
constexpr double ev2erg = 1.602176634e-12;
constexpr double MeV2eV = 1.0e6;
constexpr double MeV2erg = MeV2eV * ev2erg;
constexpr double parsec = 3.085677581467192e18;
constexpr double AU = 1.495978707e13;
constexpr double M_solar = 1.98841e33;
constexpr double R_solar = 6.957e10;
constexpr double L_solar = 3.828e33;
} // namespace C

using Real = amrex::Real;

// reference src/math/math_impl.hpp:15-18
AMREX_GPU_HOST_DEVICE inline auto clamp(double v, double lo, double hi) -> double { return (v < lo) ? lo : (hi < v) ? hi : v; }
template <typename T> AMREX_GPU_HOST_DEVICE constexpr auto sgn(T val) -> int { return (T(0) < val) - (val < T(0)); }

struct Physics_NumVars { // reference src/physics_numVars.hpp
	static const int numHydroVars = 6;
	static const int numRadVars = 4;
	// face-centred (declarations only: no face-centred state is evolved by this build — MHD is out of scope)
	static const int numMHDVars_per_dim = 1;
	static const int numVelVars_per_dim = 1;
	static const int numMHDVars_tot = AMREX_SPACEDIM * numMHDVars_per_dim;
	static const int numVelVars_tot = AMREX_SPACEDIM * numVelVars_per_dim;
};

template <typename problem_t> struct Physics_Traits {
	static constexpr bool is_hydro_enabled = false;
	static constexpr int numMassScalars = 0;
	static constexpr int numPassiveScalars = numMassScalars + 0;
	static constexpr bool is_radiation_enabled = false;
	static constexpr bool is_mhd_enabled = false;
	static constexpr int nGroups = 1;
};

template <typename problem_t> struct Physics_Indices {
	// reference src/physics_info.hpp:20-38: neither hydro nor radiation -> the single variable of an advection problem
	static constexpr int nvarTotal_cc_adv = 1;
	static constexpr int nvarTotal_cc_radhydro = []() constexpr {
		if constexpr (Physics_Traits<problem_t>::is_radiation_enabled) { // (nGroups is only read where radiation is on: advection problems do not define it)
			return Physics_Traits<problem_t>::numPassiveScalars + Physics_NumVars::numHydroVars + Physics_NumVars::numRadVars * Physics_Traits<problem_t>::nGroups;
		} else if constexpr (Physics_Traits<problem_t>::is_hydro_enabled) {
			return Physics_Traits<problem_t>::numPassiveScalars + Physics_NumVars::numHydroVars;
		} else {
			return 0;
		}
	}();
	static constexpr int nvarTotal_cc = nvarTotal_cc_radhydro > 0 ? nvarTotal_cc_radhydro : nvarTotal_cc_adv;
	static const int hydroFirstIndex = 0;
	static const int pscalarFirstIndex = Physics_NumVars::numHydroVars;
	static const int radFirstIndex = pscalarFirstIndex + Physics_Traits<problem_t>::numPassiveScalars;
	// face-centred (reference src/physics_info.hpp:43-49; declarations, see Physics_NumVars)
	static const int nvarPerDim_fc = Physics_NumVars::numVelVars_per_dim * static_cast<int>(Physics_Traits<problem_t>::is_hydro_enabled) +
					 Physics_NumVars::numMHDVars_per_dim * static_cast<int>(Physics_Traits<problem_t>::is_mhd_enabled);
	static const int nvarTotal_fc = AMREX_SPACEDIM * nvarPerDim_fc;
	static const int velFirstIndex = 0;
	static const int mhdFirstIndex = velFirstIndex + Physics_NumVars::numVelVars_per_dim;
};

namespace qkhost
{
// Problems whose face-centred state this host carries: those with the MHD index bookkeeping (FCQuantities).  The reference allocates one face
// velocity per direction for every hydro problem as well (for tracer particles; zero unless do_tracers, written to every plotfile as
// x/y/z-velocity and to every checkpoint as Level_<l>/Face_*): those zero-valued arrays are not carried here.
template <typename problem_t> constexpr auto hasFaceState() -> bool
{
	return Physics_Indices<problem_t>::nvarTotal_fc > 0 && Physics_Traits<problem_t>::is_mhd_enabled;
}
} // namespace qkhost

// reference src/hydro/mhd_system.hpp: the index bookkeeping of the face-centred magnetic field (nothing else exists there either)
template <typename problem_t> class MHDSystem
{
      public:
	static constexpr int nvar_per_dim_ = Physics_NumVars::numMHDVars_per_dim;
	static constexpr int nvar_tot_ = Physics_NumVars::numMHDVars_tot;
	enum varIndex_perDim {
		bfield_index = Physics_Indices<problem_t>::mhdFirstIndex,
	};
};

namespace quokka
{
template <typename problem_t> struct EOS_Traits {
	static constexpr double gamma = 5. / 3.;
	static constexpr double cs_isothermal = std::numeric_limits<double>::quiet_NaN();
	static constexpr double mean_molecular_weight = std::numeric_limits<double>::quiet_NaN();
	static constexpr double boltzmann_constant = C::k_B;
};
// quokka::EOS<problem_t> (reference src/hydro/EOS.hpp:40-244), host side, for problem generators: the direct gamma-law forms the
// kernels use (p = (gamma - 1) rho e, e = p / ((gamma - 1) rho); DESIGN.md section 4 on the un-vendored Microphysics EOS)
template <typename problem_t> struct EOS {
	static constexpr int nmscalars_ = Physics_Traits<problem_t>::numMassScalars;
	using MassScalars = std::optional<amrex::GpuArray<amrex::Real, nmscalars_>>; // EOS.hpp:45-66 (the gamma-law EOS ignores them)
	static constexpr double gamma_ = EOS_Traits<problem_t>::gamma;
	static constexpr double mu_ = EOS_Traits<problem_t>::mean_molecular_weight / C::m_u;
	static constexpr double kB_ = EOS_Traits<problem_t>::boltzmann_constant;
	AMREX_GPU_HOST_DEVICE static auto ComputeEintFromPres(double rho, double Pressure, MassScalars const & /*massScalars*/ = {}) -> double
	{
		double const e = Pressure / ((gamma_ - 1.0) * rho);
		return e * rho;
	}
	AMREX_GPU_HOST_DEVICE static auto ComputePressure(double rho, double Eint, MassScalars const & /*massScalars*/ = {}) -> double
	{
		double const e = Eint / rho;
		return ((gamma_ - 1.0) * rho * e) * kB_ / C::k_B;
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeSoundSpeed(double rho, double Pressure, MassScalars const & /*massScalars*/ = {}) -> double
	{
		return std::sqrt(gamma_ * Pressure / rho); // EOS.hpp:143-175 for the gamma law
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeTgasFromEint(double rho, double Eint, MassScalars const & /*massScalars*/ = {}) -> double
	{
		double const e = Eint / rho;
		return (e * mu_ * C::m_u * (gamma_ - 1.0) / C::k_B) * C::k_B / kB_;
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeEintFromTgas(double rho, double Tgas, MassScalars const & /*massScalars*/ = {}) -> double
	{
		return gammaLawEintFromTgas(rho, Tgas);
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeEintTempDerivative(double rho, double Tgas, MassScalars const & /*massScalars*/ = {}) -> double
	{
		double const p = rho * Tgas * C::k_B / (mu_ * C::m_u);
		double const e = p / ((gamma_ - 1.0) * rho);
		return (e / Tgas) * rho * kB_ / C::k_B;
	}
	// (not a hook: what ComputeEintFromTgas is unless a problem specialises it — qkhost::traits() tells the two apart with it)
	AMREX_GPU_HOST_DEVICE static auto gammaLawEintFromTgas(double rho, double Tgas) -> double
	{
		double const p = rho * Tgas * C::k_B / (mu_ * C::m_u);
		double const e = p / ((gamma_ - 1.0) * rho);
		return e * rho * kB_ / C::k_B;
	}
};
enum class direction { na = -1, x, y, z };
enum class centering { cc = 0, fc, ec };
// reference src/grid.hpp
struct grid {
	amrex::Array4<double> array_;
	amrex::Box indexRange_;
	amrex::GpuArray<double, AMREX_SPACEDIM> dx_, prob_lo_, prob_hi_;
	centering cen_ = centering::cc;
	direction dir_ = direction::na;
};
} // namespace quokka

template <typename problem_t> struct HydroSystem_Traits {
	static constexpr bool reconstruct_eint = true;
};

enum class FluxDir { X1 = 0, X2 = 1, X3 = 2 };
enum SlopeLimiter { minmod = 0, MC };
enum class RiemannSolver { HLLC, LLF, HLLD };

// process-wide C-ABI handles (amrex::Initialize analogue)
namespace qkhost
{
struct Runtime {
	qk_ctx *ctx = nullptr;
	qk_level *lev = nullptr; // the level the static operators act on (every simulation object activates its own before it launches)
	// The compute stream of the ghost fill and the fused stages: the legacy default stream, the one the problem files' ParallelFor lambdas, the
	// reference-shaped operators and the level machinery (interpolation, flux registers, average-down) run on.  It used to be a BLOCKING stream of
	// its own, which the default stream orders itself against in both directions — correct, but every hand-over between the two is a dependency
	// across two hardware queues: 15–45 us of idle GPU before each k_interp / k_physbc / k_fluxreg / memset of a refined level's step (1 900 such
	// gaps, 60 ms of a 630 ms config-5 run: profiles/tools/gpu_gaps.py).  QK_OWN_COMPUTE_STREAM=1 restores the separate stream (A/B runs).  The
	// communication stream of qk_comm.hpp is non-blocking and is ordered against this one by events only (exchangeBegin / exchangeEnd).
	hipStream_t compute = nullptr;
	bool computeChosen = false;
	auto computeStream() -> hipStream_t
	{
		if (!computeChosen) {
			computeChosen = true;
			char const *e = std::getenv("QK_OWN_COMPUTE_STREAM");
			if (e != nullptr && std::atoi(e) != 0 && hipStreamCreate(&compute) != hipSuccess) {
				amrex::Abort("hipStreamCreate (compute stream) failed");
			}
		}
		return compute;
	}
	static auto get() -> Runtime &
	{
		static Runtime r;
		return r;
	}
};
inline void check(int rc, const char *what)
{
	if (rc != QK_OK) {
		amrex::Abort(std::string(what) + ": " + qk_last_error(Runtime::get().ctx));
	}
}
inline auto tab(amrex::MultiFab const &mf) -> qk_array4 * { return reinterpret_cast<qk_array4 *>(mf.arrays()); }
inline auto itab(amrex::iMultiFab const &mf) -> qk_iarray4 * { return reinterpret_cast<qk_iarray4 *>(mf.arrays()); }
// The temperature hooks of quokka::EOS<problem_t> (reference src/hydro/EOS.hpp:74-244) run on the device in the reference.  A problem that did
// not specialise them — or specialised them to the Su & Olson material E_int = alpha / 4 T^4 — is recognised on probe points and served by the
// library's own arithmetic (qk_hydro_traits::eos_temperature_model 0 / 1: shared reciprocals, bit-identical to the CPU oracle); any other
// specialisation is compiled into the source-term kernel of the problem's translation unit (QK_HOOK_COMPILED, qk_problem_kernels.hpp).
template <typename problem_t> auto eosTemperatureModel() -> std::pair<int, double>
{
	using E = quokka::EOS<problem_t>;
	const double rs[3] = {1.0, 7.0, 2.0e-7}, Ts[3] = {1.0, 2.0, 3.0e3};
	bool gammaLaw = true, fourth = true;
	double const alpha = 4.0 * E::ComputeEintFromTgas(rs[0], Ts[0]);
	for (double r : rs) {
		for (double T : Ts) {
			double const e = E::ComputeEintFromTgas(r, T);
			gammaLaw = gammaLaw && (e == E::gammaLawEintFromTgas(r, T) || (std::isnan(e) && std::isnan(E::gammaLawEintFromTgas(r, T))));
			double const want = (alpha / 4.0) * std::pow(T, 4);
			fourth = fourth && std::abs(e - want) <= 1e-14 * std::abs(want) && std::abs(E::ComputeEintTempDerivative(r, T) - alpha * std::pow(T, 3)) <= 1e-14 * alpha * std::pow(T, 3) &&
				 std::abs(E::ComputeTgasFromEint(r, e) - T) <= 1e-13 * T;
		}
	}
	if (gammaLaw) {
		return {0, 0.0};
	}
	if (fourth && alpha > 0.0) {
		return {1, alpha};
	}
	// anything else: the problem's compiled hooks (qk_problem_kernels.hpp); the library entry points that would have to evaluate them refuse
	return {QK_HOOK_COMPILED, 0.0};
}
// members a problem's EOS_Traits specialisation may leave out (the reference only reads them in the branches that need them)
template <typename T, typename = void> struct CsIsoOf {
	static constexpr double value = std::numeric_limits<double>::quiet_NaN();
};
template <typename T> struct CsIsoOf<T, std::void_t<decltype(T::cs_isothermal)>> {
	static constexpr double value = T::cs_isothermal;
};
template <typename T, typename = void> struct MuOf {
	static constexpr double value = std::numeric_limits<double>::quiet_NaN();
};
template <typename T> struct MuOf<T, std::void_t<decltype(T::mean_molecular_weight)>> {
	static constexpr double value = T::mean_molecular_weight;
};
template <typename T, typename = void> struct KbOf {
	static constexpr double value = C::k_B;
};
template <typename T> struct KbOf<T, std::void_t<decltype(T::boltzmann_constant)>> {
	static constexpr double value = T::boltzmann_constant;
};
template <typename problem_t> auto traits() -> qk_hydro_traits
{
	return {quokka::EOS_Traits<problem_t>::gamma,
		CsIsoOf<quokka::EOS_Traits<problem_t>>::value,
		MuOf<quokka::EOS_Traits<problem_t>>::value,
		KbOf<quokka::EOS_Traits<problem_t>>::value,
		HydroSystem_Traits<problem_t>::reconstruct_eint ? 1 : 0,
		Physics_Traits<problem_t>::numPassiveScalars,
		Physics_Traits<problem_t>::numMassScalars,
		AMREX_SPACEDIM,
		eosTemperatureModel<problem_t>().first,
		eosTemperatureModel<problem_t>().second};
}
} // namespace qkhost

template <typename problem_t> class HyperbolicSystem
{
      public:
	template <FluxDir DIR> static void ReconstructStatesConstant(amrex::MultiFab const &q, amrex::MultiFab &l, amrex::MultiFab &r, int nghost, int nvars)
	{
		qkhost::check(qk_ReconstructStatesConstant(qkhost::Runtime::get().lev, nullptr, static_cast<int>(DIR), qkhost::tab(q), qkhost::tab(l), qkhost::tab(r),
							   nghost, nvars),
			      "ReconstructStatesConstant");
	}
	template <FluxDir DIR, SlopeLimiter limiter>
	static void ReconstructStatesPLM(amrex::MultiFab const &q, amrex::MultiFab &l, amrex::MultiFab &r, int nghost, int nvars)
	{
		qkhost::check(qk_ReconstructStatesPLM(qkhost::Runtime::get().lev, nullptr, static_cast<int>(DIR), static_cast<int>(limiter), qkhost::tab(q),
						      qkhost::tab(l), qkhost::tab(r), nghost, nvars),
			      "ReconstructStatesPLM");
	}
	template <FluxDir DIR>
	static void ReconstructStatesPPM(amrex::MultiFab const &q, amrex::MultiFab &l, amrex::MultiFab &r, int nghost, int nvars, int iReadFrom = 0,
					 int iWriteFrom = 0)
	{
		qkhost::check(qk_ReconstructStatesPPM(qkhost::Runtime::get().lev, nullptr, static_cast<int>(DIR), qkhost::tab(q), qkhost::tab(l), qkhost::tab(r), nghost,
						      nvars, iReadFrom, iWriteFrom),
			      "ReconstructStatesPPM");
	}
};

template <typename problem_t> class HydroSystem : public HyperbolicSystem<problem_t>
{
      public:
	static constexpr int nmscalars_ = Physics_Traits<problem_t>::numMassScalars;
	static constexpr int nscalars_ = Physics_Traits<problem_t>::numPassiveScalars;
	static constexpr int nvar_ = Physics_NumVars::numHydroVars + nscalars_;
	enum consVarIndex { density_index = 0, x1Momentum_index, x2Momentum_index, x3Momentum_index, energy_index, internalEnergy_index, scalar0_index };
	enum primVarIndex { primDensity_index = 0, x1Velocity_index, x2Velocity_index, x3Velocity_index, pressure_index, primEint_index, primScalar0_index };
	static constexpr double gamma_ = quokka::EOS_Traits<problem_t>::gamma;
	static constexpr bool reconstruct_eint = HydroSystem_Traits<problem_t>::reconstruct_eint;

	static auto lev() -> qk_level * { return qkhost::Runtime::get().lev; }

	// per-cell functions problems call inside their own device lambdas (ErrorEst, diagnostics): hydro_system.hpp:349-394, with the direct
	// gamma-law forms of the library (qk_device.hpp consPressure / Eos::soundSpeed)
	AMREX_GPU_HOST_DEVICE static auto ComputePressure(amrex::Array4<const amrex::Real> const &cons, int i, int j, int k) -> amrex::Real
	{
		const auto rho = cons(i, j, k, density_index);
		if constexpr (gamma_ == 1.0) {
			return rho * qkhost::CsIsoOf<quokka::EOS_Traits<problem_t>>::value * qkhost::CsIsoOf<quokka::EOS_Traits<problem_t>>::value;
		}
		const auto vx = cons(i, j, k, x1Momentum_index) / rho;
		const auto vy = cons(i, j, k, x2Momentum_index) / rho;
		const auto vz = cons(i, j, k, x3Momentum_index) / rho;
		const auto kinetic_energy = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
		const auto thermal_energy = cons(i, j, k, energy_index) - kinetic_energy;
		const auto e = (rho == 0.0) ? 0.0 : thermal_energy / rho;
		return (gamma_ - 1.0) * rho * e;
	}
	// hydro_system.hpp:294-347: primitive <-> conserved state of one cell (boundary functors: NSCBC)
	AMREX_GPU_HOST_DEVICE static auto ComputePrimVars(amrex::Array4<const amrex::Real> const &cons, int i, int j, int k) -> quokka::valarray<amrex::Real, nvar_>
	{
		const auto rho = cons(i, j, k, density_index);
		const auto vx = cons(i, j, k, x1Momentum_index) / rho;
		const auto vy = cons(i, j, k, x2Momentum_index) / rho;
		const auto vz = cons(i, j, k, x3Momentum_index) / rho;
		quokka::valarray<amrex::Real, nvar_> primVars{rho, vx, vy, vz, ComputePressure(cons, i, j, k), cons(i, j, k, internalEnergy_index)};
		for (int n = 0; n < nscalars_; ++n) {
			primVars[primScalar0_index + n] = cons(i, j, k, scalar0_index + n);
		}
		return primVars;
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeConsVars(quokka::valarray<amrex::Real, nvar_> const &prim) -> quokka::valarray<amrex::Real, nvar_>
	{
		amrex::Real const rho = prim[0], v1 = prim[1], v2 = prim[2], v3 = prim[3];
		amrex::Real const Eint = quokka::EOS<problem_t>::ComputeEintFromPres(rho, prim[4]);
		amrex::Real const Egas = Eint + 0.5 * rho * (v1 * v1 + v2 * v2 + v3 * v3);
		quokka::valarray<amrex::Real, nvar_> consVars{rho, rho * v1, rho * v2, rho * v3, Egas, prim[5]};
		for (int n = 0; n < nscalars_; ++n) {
			consVars[scalar0_index + n] = prim[primScalar0_index + n];
		}
		return consVars;
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeSoundSpeed(amrex::Array4<const amrex::Real> const &cons, int i, int j, int k) -> amrex::Real
	{
		if constexpr (gamma_ == 1.0) {
			return qkhost::CsIsoOf<quokka::EOS_Traits<problem_t>>::value;
		}
		return std::sqrt(gamma_ * ComputePressure(cons, i, j, k) / cons(i, j, k, density_index));
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeVelocityX1(amrex::Array4<const amrex::Real> const &cons, int i, int j, int k) -> amrex::Real
	{
		return cons(i, j, k, x1Momentum_index) / cons(i, j, k, density_index);
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeVelocityX2(amrex::Array4<const amrex::Real> const &cons, int i, int j, int k) -> amrex::Real
	{
		return cons(i, j, k, x2Momentum_index) / cons(i, j, k, density_index);
	}
	AMREX_GPU_HOST_DEVICE static auto ComputeVelocityX3(amrex::Array4<const amrex::Real> const &cons, int i, int j, int k) -> amrex::Real
	{
		return cons(i, j, k, x3Momentum_index) / cons(i, j, k, density_index);
	}

	static void ConservedToPrimitive(amrex::MultiFab const &cons, amrex::MultiFab &prim, int nghost)
	{
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_hydro_ConservedToPrimitive(lev(), nullptr, &t, qkhost::tab(cons), qkhost::tab(prim), nghost), "ConservedToPrimitive");
	}
	template <FluxDir DIR> static void ComputeFlatteningCoefficients(amrex::MultiFab const &prim, amrex::MultiFab &chi, int nghost)
	{
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_hydro_ComputeFlatteningCoefficients(lev(), nullptr, &t, static_cast<int>(DIR), qkhost::tab(prim), qkhost::tab(chi), nghost),
			      "ComputeFlatteningCoefficients");
	}
	template <FluxDir DIR>
	static void FlattenShocks(amrex::MultiFab const &q, amrex::MultiFab const &c1, amrex::MultiFab const &c2, amrex::MultiFab const &c3, amrex::MultiFab &l,
				  amrex::MultiFab &r, int nghost, int nvars)
	{
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_hydro_FlattenShocks(lev(), nullptr, &t, static_cast<int>(DIR), qkhost::tab(q), qkhost::tab(c1), qkhost::tab(c2), qkhost::tab(c3),
						     qkhost::tab(l), qkhost::tab(r), nghost, nvars),
			      "FlattenShocks");
	}
	template <RiemannSolver RIEMANN, FluxDir DIR>
	static void ComputeFluxes(amrex::MultiFab &flux, amrex::MultiFab &fvel, amrex::MultiFab const &l, amrex::MultiFab const &r, amrex::MultiFab const &prim,
				  amrex::Real K_visc)
	{
		// (HLLD: the reference's MHD stub — zero magnetic field, hydro_system.hpp:987-1003, :1044-1048)
		constexpr int riemann = (RIEMANN == RiemannSolver::LLF) ? QK_RIEMANN_LLF : (RIEMANN == RiemannSolver::HLLD) ? QK_RIEMANN_HLLD : QK_RIEMANN_HLLC;
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_hydro_ComputeFluxes(lev(), nullptr, &t, riemann, static_cast<int>(DIR),
						     qkhost::tab(flux), qkhost::tab(fvel), qkhost::tab(l), qkhost::tab(r), qkhost::tab(prim), K_visc),
			      "ComputeFluxes");
	}
	static void ComputeRhsFromFluxes(amrex::MultiFab &rhs, std::array<amrex::MultiFab, AMREX_SPACEDIM> const &fluxArray,
					 amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> dx, int nvars)
	{
		auto t = qkhost::traits<problem_t>();
		const qk_array4 *f[3] = {nullptr, nullptr, nullptr};
		double d3[3] = {1, 1, 1};
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			f[d] = qkhost::tab(fluxArray[d]);
			d3[d] = dx[d];
		}
		qkhost::check(qk_hydro_ComputeRhsFromFluxes(lev(), nullptr, &t, qkhost::tab(rhs), f, d3, nvars), "ComputeRhsFromFluxes");
	}
	static void AddInternalEnergyPdV(amrex::MultiFab &rhs, amrex::MultiFab const &cons, amrex::GpuArray<amrex::Real, AMREX_SPACEDIM> dx,
					 std::array<amrex::MultiFab, AMREX_SPACEDIM> const &faceVel, amrex::iMultiFab const &redoFlag)
	{
		auto t = qkhost::traits<problem_t>();
		const qk_array4 *v[3] = {nullptr, nullptr, nullptr};
		double d3[3] = {1, 1, 1};
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			v[d] = qkhost::tab(faceVel[d]);
			d3[d] = dx[d];
		}
		qkhost::check(qk_hydro_AddInternalEnergyPdV(lev(), nullptr, &t, qkhost::tab(rhs), qkhost::tab(cons), d3, v, qkhost::itab(redoFlag)),
			      "AddInternalEnergyPdV");
	}
	static void PredictStep(amrex::MultiFab const &old, amrex::MultiFab &neu, amrex::MultiFab const &rhs, double dt, int nvars, amrex::iMultiFab &redoFlag,
				int64_t *d_redo_count = nullptr)
	{
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_hydro_PredictStep(lev(), nullptr, &t, qkhost::tab(old), qkhost::tab(neu), qkhost::tab(rhs), dt, nvars, qkhost::itab(redoFlag),
						   d_redo_count),
			      "PredictStep");
	}
	static void EnforceLimits(amrex::Real densityFloor, amrex::Real tempFloor, amrex::MultiFab &state)
	{
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_hydro_EnforceLimits(lev(), nullptr, &t, densityFloor, tempFloor, qkhost::tab(state)), "EnforceLimits");
	}
	static void SyncDualEnergy(amrex::MultiFab &cons, int *d_error_flag = nullptr)
	{
		auto t = qkhost::traits<problem_t>();
		qkhost::check(qk_hydro_SyncDualEnergy(lev(), nullptr, &t, qkhost::tab(cons), d_error_flag), "SyncDualEnergy");
	}
	// ParReduce max over the local valid cells (result on the host)
	static auto maxSignalSpeedLocal(amrex::MultiFab const &cons, int which = 0) -> amrex::Real
	{
		auto t = qkhost::traits<problem_t>();
		static double *d_res = nullptr;
		if (d_res == nullptr) {
			QK_HOST_HIP(hipMalloc(reinterpret_cast<void **>(&d_res), sizeof(double)));
		}
		qkhost::check(qk_hydro_maxSignalSpeedLocal(lev(), nullptr, &t, which, qkhost::tab(cons), d_res), "maxSignalSpeedLocal");
		double h = 0;
		QK_HOST_HIP(hipMemcpy(&h, d_res, sizeof(double), hipMemcpyDeviceToHost));
		return qkhost::Comm::get().allReduceMax(h); // ParallelDescriptor::ReduceRealMax (reference src/simulation.hpp:1003)
	}
};


#endif // QK_HOST_QUOKKA_HYDRO_SYSTEM_HPP_
