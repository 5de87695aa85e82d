// microphysics_stub.hpp — what the reference's problem files call from the AMReX-Astro Microphysics submodule (un-vendored: psharda/Microphysics,
// `.gitmodules:10-13`) before they touch quokka::EOS: the run-time initialisation of its parameter and EOS tables.  The gamma-law EOS of this host
// mirror (quokka::EOS<problem_t>, quokka_host.hpp) has no tables: these are the same entry points with nothing to do.  The burner / network / table
// interfaces the chemistry and tabulated-EOS problems use beyond them are NOT provided (those problems are outside the hot path).
#ifndef QK_HOST_COMPAT_MICROPHYSICS_STUB_HPP_
#define QK_HOST_COMPAT_MICROPHYSICS_STUB_HPP_

inline void init_extern_parameters() {}
inline void eos_init(double /*small_temp*/ = 0.0, double /*small_dens*/ = 0.0) {}

#endif // QK_HOST_COMPAT_MICROPHYSICS_STUB_HPP_
