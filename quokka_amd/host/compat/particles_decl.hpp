// particles_decl.hpp — DECLARATIONS of the particle types the reference's problem files name (src/particles/CICParticles.hpp over
// amrex::AmrParticleContainer, AMReX_Particles.H), so that those files compile and link unchanged.  Particles — cloud-in-cell deposition for the
// Poisson solve, tracers — are outside this path and are not built: a container cannot be created (AMRSimulation never makes one,
// `CICParticles` stays null), and every member that would touch particle data refuses.
#ifndef QK_HOST_COMPAT_PARTICLES_DECL_HPP_
#define QK_HOST_COMPAT_PARTICLES_DECL_HPP_

#include <array>
#include <vector>

#include "../amrex_mini.hpp"

namespace amrex
{
using ParticleReal = Real;

template <int NReal, int NInt = 0> struct Particle {
	Real m_pos[AMREX_SPACEDIM] = {};
	Real m_rdata[NReal > 0 ? NReal : 1] = {};
	int m_idata[NInt > 0 ? NInt : 1] = {};
	long m_id = 0;
	int m_cpu = 0;
	QK_HD auto pos(int d) -> Real & { return m_pos[d]; }
	QK_HD auto pos(int d) const -> Real const & { return m_pos[d]; }
	QK_HD auto rdata(int n) -> Real & { return m_rdata[n]; }
	QK_HD auto rdata(int n) const -> Real const & { return m_rdata[n]; }
	QK_HD auto idata(int n) -> int & { return m_idata[n]; }
	auto id() -> long & { return m_id; }
	auto cpu() -> int & { return m_cpu; }
	static auto NextID() -> long
	{
		static long n = 0;
		return ++n;
	}
};

template <int NReal, int NInt = 0, int NArrayReal = 0, int NArrayInt = 0> class ParticleContainer
{
      public:
	using ParticleType = Particle<NReal, NInt>;
	struct ParticleInitData {
		std::array<double, NReal> real_struct_data{};
		std::array<int, NInt> int_struct_data{};
		std::array<double, NArrayReal> real_array_data{};
		std::array<int, NArrayInt> int_array_data{};
	};
	using ParticleTileType = std::vector<ParticleType>;
	// the array of structs of one tile as AMReX hands it out: aos()() is the vector of particles
	struct ArrayOfStructs {
		std::vector<ParticleType> v;
		auto operator()() -> std::vector<ParticleType> & { return v; }
		auto operator()() const -> std::vector<ParticleType> const & { return v; }
	};
	[[noreturn]] static void notBuilt() { Abort("amrex::ParticleContainer: particles are not built in quokka_amd/host (declarations only)"); }
	template <typename G, typename D, typename B> void Define(G const & /*geom*/, D const & /*dmap*/, B const & /*ba*/) { notBuilt(); }
	template <typename PC> void copyParticles(PC const & /*other*/, bool /*local*/ = false) { notBuilt(); }
	void InitRandom(Long /*icount*/, unsigned long /*iseed*/, ParticleInitData const & /*pdata*/, bool /*serialize*/ = false) { notBuilt(); }
	void InitOnePerCell(Real /*x*/, Real /*y*/, Real /*z*/, ParticleInitData const & /*pdata*/) { notBuilt(); }
	void InitFromAsciiFile(std::string const & /*file*/, int /*extradata*/, const IntVect * /*Nrep*/ = nullptr) { notBuilt(); }
	void Redistribute() { notBuilt(); }
	void SetVerbose(int /*v*/) {}
	auto DefineAndReturnParticleTile(int /*lev*/, int /*grid*/, int /*tile*/) -> ParticleTileType &
	{
		notBuilt();
		static ParticleTileType t;
		return t;
	}
	[[nodiscard]] auto TotalNumberOfParticles() const -> Long { return 0; }
	[[nodiscard]] auto finestLevel() const -> int { return 0; }
};
template <int NReal, int NInt = 0, int NArrayReal = 0, int NArrayInt = 0> using AmrParticleContainer = ParticleContainer<NReal, NInt, NArrayReal, NArrayInt>;

// amrex::ParIter over the particle tiles of a level: there are none
template <int NReal, int NInt = 0> class ParIter
{
      public:
	using Container = ParticleContainer<NReal, NInt>;
	ParIter(Container const & /*pc*/, int /*lev*/) {}
	[[nodiscard]] auto isValid() const -> bool { return false; }
	void operator++() {}
	[[nodiscard]] auto numParticles() const -> Long { return 0; }
	[[nodiscard]] auto GetArrayOfStructs() const -> typename Container::ArrayOfStructs &
	{
		static typename Container::ArrayOfStructs a;
		return a;
	}
};
} // namespace amrex

namespace quokka
{
enum ParticleDataIdx { ParticleMassIdx = 0, ParticleVxIdx, ParticleVyIdx, ParticleVzIdx };
constexpr int CICParticleRealComps = 4; // mass vx vy vz
using CICParticleContainer = amrex::AmrParticleContainer<CICParticleRealComps>;
using CICParticleIterator = amrex::ParIter<CICParticleRealComps>;
} // namespace quokka

#endif // QK_HOST_COMPAT_PARTICLES_DECL_HPP_
