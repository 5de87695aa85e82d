/* quokka_amd.h — C-ABI of the MI355X-native hydro / radiation hot path.
 *
 * Drop-in boundary for the reference's operator surface.  The reference has no FFI:
 * its "operator API" is a set of static C++ template methods of HydroSystem<problem_t>,
 * HyperbolicSystem<problem_t>, RadSystem<problem_t> taking AMReX MultiFabs, plus the level-0 branch
 * of AMRSimulation::fillBoundaryConditions.  Every entry point below replaces exactly one of those
 * methods (cited as `reference file:line`), takes plain pointers and sizes, never throws, returns
 * 0 on success or a negative qk_status, allocates nothing the caller did not hand over (scratch is
 * caller-provided), is asynchronous on the supplied HIP stream and keeps no global mutable state
 * outside the opaque qk_ctx.
 *
 * Data contract (== amrex::MultiFab on device):
 *   - one FP64 array per box, Fortran order, component index outermost, ghost cells included;
 *   - a MultiFab argument is a DEVICE pointer to `nboxes` qk_array4 descriptors.  qk_array4 is
 *     binary-compatible with amrex::Array4<double> (AMReX_Array4.H: p, jstride, kstride, nstride,
 *     begin, end, ncomp), so `mf.arrays()` (a device MultiArray4) can be passed as is;
 *   - launch geometry comes from a qk_level (the BoxArray of valid boxes), built once.
 *
 * All kernels are compiled with -ffp-contract=off and keep the reference's association order
 * (reference CMakeLists.txt:31 DISABLE_FMAD=ON).
 */
#ifndef QUOKKA_AMD_H_
#define QUOKKA_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qk_ctx qk_ctx;
typedef struct qk_level qk_level;
typedef void *qk_stream; /* hipStream_t */

typedef enum qk_status {
	QK_OK = 0,
	QK_ERR_INVALID = -1,	 /* bad argument (null pointer, unknown enum, mismatched sizes) */
	QK_ERR_HIP = -2,	 /* a HIP runtime call failed; see qk_last_error() */
	QK_ERR_UNSUPPORTED = -3, /* valid reference feature this build does not cover (e.g. mass scalars) */
	QK_ERR_STATE = -4	 /* physical state error the reference treats as fatal (rho <= 0 in SyncDualEnergy) */
} qk_status;

/* == amrex::Array4<double> */
typedef struct qk_array4 {
	double *p;
	int64_t jstride, kstride, nstride;
	int begin[3]; /* inclusive lower corner (with ghosts) */
	int end[3];   /* exclusive upper corner (hi + 1) */
	int ncomp;
} qk_array4;

/* == amrex::Array4<int> (iMultiFab redoFlag) */
typedef struct qk_iarray4 {
	int *p;
	int64_t jstride, kstride, nstride;
	int begin[3];
	int end[3];
	int ncomp;
} qk_iarray4;

typedef struct qk_box {
	int lo[3];
	int hi[3]; /* inclusive */
} qk_box;

/* compile-time traits of the reference, as run-time data:
 * quokka::EOS_Traits<P> (src/hydro/EOS.hpp:32-37), HydroSystem_Traits<P> (src/hydro/hydro_system.hpp:38-41),
 * Physics_Traits<P> (src/physics_info.hpp:8-17), AMREX_SPACEDIM. */
#define QK_MAX_SCALARS 8
/* value of qk_hydro_traits::eos_temperature_model, qk_rad_traits::opacity_model and qk_rad_traits::thermal_model that stands for "this hook
 * is the problem's own compiled device function": only meaningful to the kernels a problem's translation unit instantiates itself
 * (quokka_amd/host/qk_problem_kernels.hpp); every entry point of this library that would have to EVALUATE such a hook returns
 * QK_ERR_UNSUPPORTED, the others ignore the field. */
#define QK_HOOK_COMPILED 100
typedef struct qk_hydro_traits {
	double gamma;
	double cs_isothermal;
	double mean_molecular_weight;
	double boltzmann_constant;
	int reconstruct_eint;
	int nscalars;  /* Physics_Traits::numPassiveScalars, 0..QK_MAX_SCALARS: carried by the reference-shaped operators (state / flux arrays hold
			* 6 + nscalars components); the fused stage carries up to 3 passive scalars and refuses more, or mass scalars (QK_ERR_UNSUPPORTED) */
	int nmscalars; /* Physics_Traits::numMassScalars, 0..nscalars: the first nmscalars passive scalars are partial densities — consistent
			* multi-fluid advection of their fluxes (hydro_system.hpp:1062-1073,1094-1104), non-negativity in isStateValid (:430-441),
			* floor and renormalisation in EnforceLimits (:725-744; small_x = 1e-30) */
	int ndim;      /* AMREX_SPACEDIM of the build: 1, 2 (X2 = the index-swap view of ArrayView_2d.hpp) or 3 */
	/* the quokka::EOS<P> temperature hooks a problem may specialise (ComputeTgasFromEint / ComputeEintFromTgas / ComputeEintTempDerivative,
	 * reference src/hydro/EOS.hpp:74-244), closed set used by the radiation source terms:
	 * 0: the gamma-law forms; 1: E_int = (eos_alpha / 4) T^4 (Su & Olson 1997; RadMatterCoupling, RadSuOlson, RadMarshak) */
	int eos_temperature_model;
	double eos_alpha;
} qk_hydro_traits;

enum { QK_DIR_X1 = 0, QK_DIR_X2 = 1, QK_DIR_X3 = 2 };
enum { QK_RIEMANN_HLLC = 0, QK_RIEMANN_LLF = 1, QK_RIEMANN_HLLD = 2 /* reference-shaped operator only; the MHD stub of hydro_system.hpp:987-1003: B = 0 */ };
enum { QK_LIMITER_MINMOD = 0, QK_LIMITER_MC = 1 };
/* amrex::BCType values */
enum { QK_BC_REFLECT_ODD = -1, QK_BC_INT_DIR = 0, QK_BC_REFLECT_EVEN = 1, QK_BC_FOEXTRAP = 2, QK_BC_EXT_DIR = 3 };

/* ------------------------------------------------------------------ context / level */
#define QK_DEVICE_HOST_PLANNING (-1) /* planning-only context: host box / ghost-plan logic, no kernels (multi-rank CPU tests) */
int qk_ctx_create(qk_ctx **ctx, int device);
int qk_ctx_destroy(qk_ctx *ctx);
const char *qk_last_error(qk_ctx *ctx);
const char *qk_version(void);

/* Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline
 * figure).  qk_profile_num_kernels / qk_profile_get synchronise on the recorded events. */
int qk_profile_enable(qk_ctx *ctx, int on);
/* restrict the timing to the kernel of that name (NULL or "": every kernel).  Each timed launch costs two event records on the stream (~3 us of GPU
 * time each): bench.py times its measured region with events around the dominant kernel only and fills the per-kernel table in a separate pass. */
int qk_profile_only(qk_ctx *ctx, const char *kernel_name);
int qk_profile_reset(qk_ctx *ctx);
int qk_profile_num_kernels(qk_ctx *ctx);
int qk_profile_get(qk_ctx *ctx, int k, const char **name, long *count, double *total_ms);

/* hipMemsetAsync on the caller's stream: the device words a fused stage reports in (redo count, CFL maxima) are cleared by the library's own
 * call instead of a host-framework fill kernel (the reference clears its counters with amrex::Gpu fills, e.g. redoFlag.setVal(none),
 * src/QuokkaSimulation.hpp:1087).  `device_ptr` must be a device allocation. */
int qk_clear_bytes(qk_ctx *ctx, qk_stream s, void *device_ptr, int64_t nbytes);

/* BoxArray of one level owned by this rank (valid, cell-centred boxes). */
int qk_level_create(qk_ctx *ctx, qk_level **lev, int ndim, int nboxes, const qk_box *valid_boxes);
int qk_level_destroy(qk_level *lev);

/* Convenience for callers that hold host-side descriptors: copies `n` descriptors into device
 * memory owned by the context; the returned pointer stays valid until qk_ctx_destroy. */
int qk_upload_array4_table(qk_ctx *ctx, int n, const qk_array4 *host_table, qk_array4 **device_table);
int qk_upload_iarray4_table(qk_ctx *ctx, int n, const qk_iarray4 *host_table, qk_iarray4 **device_table);

/* ------------------------------------------------------------------ HyperbolicSystem<problem_t> */
/* ReconstructStatesConstant<DIR>(q, leftState, rightState, nghost, nvars)   reference src/hyperbolic_system.hpp:129-147 */
int qk_ReconstructStatesConstant(qk_level *lev, qk_stream s, int dir, const qk_array4 *q, qk_array4 *leftState, qk_array4 *rightState, int nghost,
				 int nvars);
/* ReconstructStatesPLM<DIR,limiter>(...)                                   reference src/hyperbolic_system.hpp:183-201 */
int qk_ReconstructStatesPLM(qk_level *lev, qk_stream s, int dir, int limiter, const qk_array4 *q, qk_array4 *leftState, qk_array4 *rightState,
			    int nghost, int nvars);
/* ReconstructStatesPPM<DIR>(q, leftState, rightState, nghost, nvars, iReadFrom, iWriteFrom)   reference src/hyperbolic_system.hpp:295-316 */
int qk_ReconstructStatesPPM(qk_level *lev, qk_stream s, int dir, const qk_array4 *q, qk_array4 *leftState, qk_array4 *rightState, int nghost,
			    int nvars, int iReadFrom, int iWriteFrom);

/* ------------------------------------------------------------------ HydroSystem<problem_t> */
/* ConservedToPrimitive(cons, primVar, nghost)                               reference src/hydro/hydro_system.hpp:138-196 */
int qk_hydro_ConservedToPrimitive(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_array4 *cons, qk_array4 *primVar, int nghost);
/* ComputeFlatteningCoefficients<DIR>(primVar, x1Chi, nghost)                reference src/hydro/hydro_system.hpp:531-626 */
int qk_hydro_ComputeFlatteningCoefficients(qk_level *lev, qk_stream s, const qk_hydro_traits *t, int dir, const qk_array4 *primVar, qk_array4 *x1Chi,
					   int nghost);
/* FlattenShocks<DIR>(q, x1Chi, x2Chi, x3Chi, x1LeftState, x1RightState, nghost, nvars)   reference src/hydro/hydro_system.hpp:628-694 */
int qk_hydro_FlattenShocks(qk_level *lev, qk_stream s, const qk_hydro_traits *t, int dir, const qk_array4 *q, const qk_array4 *x1Chi,
			   const qk_array4 *x2Chi, const qk_array4 *x3Chi, qk_array4 *x1LeftState, qk_array4 *x1RightState, int nghost, int nvars);
/* ComputeFluxes<RIEMANN,DIR>(x1Flux, x1FaceVel, x1LeftState, x1RightState, primVar, K_visc)   reference src/hydro/hydro_system.hpp:852-1112
 * (HLLC: src/hydro/HLLC.hpp:22-153, LLF: src/hydro/LLF.hpp:16-43) */
int qk_hydro_ComputeFluxes(qk_level *lev, qk_stream s, const qk_hydro_traits *t, int riemann, int dir, qk_array4 *x1Flux, qk_array4 *x1FaceVel,
			   const qk_array4 *x1LeftState, const qk_array4 *x1RightState, const qk_array4 *primVar, double K_visc);
/* ComputeRhsFromFluxes(rhs, fluxArray, dx, nvars)                           reference src/hydro/hydro_system.hpp:448-473 */
int qk_hydro_ComputeRhsFromFluxes(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *rhs, const qk_array4 *const fluxArray[3],
				  const double dx[3], int nvars);
/* AddInternalEnergyPdV(rhs, consVar, dx, faceVelArray, redoFlag)            reference src/hydro/hydro_system.hpp:775-814 */
int qk_hydro_AddInternalEnergyPdV(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *rhs, const qk_array4 *consVar, const double dx[3],
				  const qk_array4 *const faceVelArray[3], const qk_iarray4 *redoFlag);
/* PredictStep(consVarOld, consVarNew, rhs, dt, nvars, redoFlag)             reference src/hydro/hydro_system.hpp:475-497
 * `d_redo_count` (device int64, may be NULL) is incremented by the number of flagged cells: it replaces the
 * separate redoFlag.sum(0) reduction of reference src/QuokkaSimulation.hpp:1146. */
int qk_hydro_PredictStep(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_array4 *consVarOld, qk_array4 *consVarNew,
			 const qk_array4 *rhs, double dt, int nvars, qk_iarray4 *redoFlag, int64_t *d_redo_count);
/* EnforceLimits(densityFloor, tempFloor, state)                             reference src/hydro/hydro_system.hpp:696-773 */
int qk_hydro_EnforceLimits(qk_level *lev, qk_stream s, const qk_hydro_traits *t, double densityFloor, double tempFloor, qk_array4 *state);
/* SyncDualEnergy(consVar)                                                   reference src/hydro/hydro_system.hpp:816-850
 * rho <= 0 is fatal in the reference (amrex::Abort); here the kernel raises `d_error_flag` (device int, may
 * be NULL) and leaves the cell untouched; the host driver turns that into QK_ERR_STATE. */
int qk_hydro_SyncDualEnergy(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *consVar, int *d_error_flag);
/* ComputeMaxSignalSpeed(cons, maxSignal, indexRange) per box                reference src/hydro/hydro_system.hpp:223-252 */
int qk_hydro_ComputeMaxSignalSpeed(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_array4 *cons, qk_array4 *maxSignal);
/* maxSignalSpeedLocal(cons) (ParReduce max over valid cells)                reference src/hydro/hydro_system.hpp:198-221
 * and max_signal_speed_.norminf()                                          reference src/simulation.hpp:710
 * which = 0: cs + sqrt(2 KE / rho) (maxSignalSpeedLocal) ; which = 1: cs + |v| (ComputeMaxSignalSpeed + norminf).
 * Result: *d_result (device double) = max over all local valid cells; deterministic (max is exact). */
int qk_hydro_maxSignalSpeedLocal(qk_level *lev, qk_stream s, const qk_hydro_traits *t, int which, const qk_array4 *cons, double *d_result);

/* ------------------------------------------------------------------ QuokkaSimulation<problem_t> helpers on the path */
/* replaceFluxes(fluxes, FOfluxes, redoFlag) for one direction               reference src/QuokkaSimulation.hpp:1324-1368
 * `face_ncomp` = nvars for fluxes, 1 for face velocities. redoFlag must have >= 1 filled ghost cell. */
int qk_replaceFluxes(qk_level *lev, qk_stream s, int dir, qk_array4 *flux, const qk_array4 *FOflux, const qk_iarray4 *redoFlag, int face_ncomp);
/* MultiFab::Saxpy(dst, a, src, 0, 0, ncomp, 0) on the face boxes of `dir` (dir = -1: cell boxes)   reference src/QuokkaSimulation.hpp:1105-1108 */
int qk_Saxpy(qk_level *lev, qk_stream s, int dir, qk_array4 *dst, double a, const qk_array4 *src, int ncomp);
/* FixupState(lev) = EnforceLimits + SyncDualEnergy (use_dual_energy == 1) in one pass, with the CFL maxima of the result — maxSignalSpeedLocal
 * which = 0 into d_max_signal[0], which = 1 into [1], cleared first; NULL: not computed      reference src/QuokkaSimulation.hpp:761-770 */
int qk_hydro_FixupState(qk_level *lev, qk_stream s, const qk_hydro_traits *t, double densityFloor, double tempFloor, int use_dual_energy, qk_array4 *state,
			int *d_error_flag, double *d_max_signal);


/* ------------------------------------------------------------------ LinearAdvectionSystem<problem_t> (reference src/linear_advection/linear_advection.hpp)
 * the scalar advection solver shares HyperbolicSystem's reconstruction (qk_ReconstructStates*); its own operators: */
/* ComputeFluxes<DIR>(x1Flux, x1LeftState, x1RightState, advectionVx, nvars): the upwind state times the velocity   linear_advection.hpp:165-198 */
int qk_advect_ComputeFluxes(qk_level *lev, qk_stream s, int dir, qk_array4 *flux, const qk_array4 *left, const qk_array4 *right, double advectionVx, int nvars);
/* PredictStep(consVarOld, consVarNew, fluxArray, dt, dx, nvars)                                                   linear_advection.hpp:82-118 */
int qk_advect_PredictStep(qk_level *lev, qk_stream s, const qk_array4 *consVarOld, qk_array4 *consVarNew, const qk_array4 *const fluxArray[3], double dt,
			  const double dx[3], int nvars);
/* AddFluxesRK2(U_new, U0, U1, fluxArray, dt, dx, nvars) (U_new may alias U1)                                       linear_advection.hpp:120-163 */
int qk_advect_AddFluxesRK2(qk_level *lev, qk_stream s, qk_array4 *U_new, const qk_array4 *U0, const qk_array4 *U1, const qk_array4 *const fluxArray[3], double dt,
			   const double dx[3], int nvars);

/* ------------------------------------------------------------------ fused fast path (MI355X design; same results) */
typedef struct qk_hydro_stage_args {
	/* inputs */
	const qk_array4 *U_in;	 /* ghost-filled state the fluxes are evaluated on (stage 1: U^n, stage 2: U^1) */
	const qk_array4 *U_old;	 /* U^n (update base and pressure for the PdV term)                       */
	qk_array4 *U_out;	 /* stage 1: U^1 ; stage 2: U^{n+1}  (valid cells only)                      */
	qk_array4 *halfFlux[3];	 /* stage 1: written with F1_d (face, nvar); stage 2: read                  */
	qk_array4 *halfVel[3];	 /* stage 1: written with v1_d (face, 1);   stage 2: read                   */
	qk_iarray4 *redoFlag;	 /* written: 0 / 1 per valid cell                                           */
	int64_t *d_redo_count;	 /* device counter, incremented by the number of flagged cells              */
	int *d_error_flag;	 /* device int, set to 1 if SyncDualEnergy meets rho <= 0                   */
	double *d_max_signal;	 /* optional device double[2] (caller zeroes): max over the valid cells of U_out of
				  * [0] cs + sqrt(2 KE / rho)  (maxSignalSpeedLocal, used by isCflViolated) and
				  * [1] cs + |v|               (ComputeMaxSignalSpeed + norminf, used by computeTimestep),
				  * folded into the epilogue so the two reduction passes over the new state disappear */
	/* scratch: caller-owned, at least qk_hydro_stage_scratch_bytes() */
	void *scratch;
	int64_t scratch_bytes;
	/* parameters */
	double dx[3];
	double dt;
	int stage;		 /* 1 or 2 */
	int reconstruction_order; /* 1, 2, 3 */
	double densityFloor, tempFloor;
	int use_dual_energy;
	double K_visc;		 /* artificial-viscosity coefficient (hydro.artificial_viscosity_coefficient, reference hydro_system.hpp:1054-1076), >= 0 */
	int store_flux_rk2;	 /* stage 2 only: write flux_rk2 = 0.5 F1 + 0.5 F2 (what the flux registers of an AMR hierarchy accumulate,
				  * reference src/QuokkaSimulation.hpp:1303-1306) into fluxRk2[d]; 0: not stored */
	qk_array4 *fluxRk2[3];	 /* face-centred like halfFlux, 6 components; required when store_flux_rk2 != 0.  Separate from halfFlux: the x
				  * sweep evaluates the face between two of its tiles in both of them, and both need the stage-1 flux intact */
	int rk2_carry_rhs;	 /* 0: flux_rk2 = 0.5 F1 + 0.5 F2 face by face, as the reference forms it (QuokkaSimulation.hpp:1106, :1220) — bit-identical
				  *    to the reference-shaped operators;
				  * 1: the RK2 average is taken on the cell instead of on the faces: stage 1 stores the HALF STEP
				  *    S = U_old + (dt / 2) rhs_1 (its own right-hand side, P dV term included) and P(U_old) in `rhs1`, stage 2 finishes
				  *    U_new = S + (dt / 2) rhs_2 with the P dV term on the stored pressure and reads neither U_old nor a second
				  *    right-hand side (dt must be the same in both stages).  U_old + dt (rhs_1 + rhs_2) / 2 in exact arithmetic, rounded
				  *    differently (~1e-16 relative per step; within the 1e-12 of the parity contract, tests/test_hydro_step_gpu.py); the face arrays halfFlux /
				  *    halfVel are neither written nor read (they may be NULL), so flux_rk2 does not exist: excludes store_flux_rk2,
				  *    and a stage-2 first-order flux correction must recompute F1 from U_old with qk_hydro_ComputeFluxes */
	qk_array4 *rhs1;	 /* rk2_carry_rhs: cell-centred, no ghost cells, 6 + nscalars + 1 components; must survive from stage 1 to stage 2 */
	const struct qk_carray4 *flux_mask; /* (qk_carray4: the byte-array descriptor declared with the AMR entries below) rk2_carry_rhs only, optional (NULL: none): cell-centred bytes with ONE ghost cell.  A face with a non-zero byte on
				  * either side is treated as the reference's form treats every face: stage 1 stores F1 in halfFlux[d], stage 2 writes
				  * flux_rk2 = 0.5 F1 + 0.5 F2 to fluxRk2[d] (both required then).  For a level with refined children: mark the coarse
				  * cells along the coarse-fine interface (the items of its flux register) and incrementFluxRegisters
				  * (src/simulation.hpp:1345-1387) finds flux_rk2 where it reads it, while the 99.9 % other faces stay carried.
				  * A box's descriptor may be a WINDOW of its array (same p-relative memory and strides, begin / end cropped to the bounding
				  * box of the marked cells; end <= begin: none): cells outside [begin, end) count as unmarked and are not read — pass
				  * windows, or every face of the level waits for two byte loads. */
	int fofc_pass;		 /* 0: the stage proper.  1: the FIRST-ORDER FLUX CORRECTION of a stage whose first pass counted flagged cells (reference
				  * src/QuokkaSimulation.hpp:1144-1184, :1232-1270), as ONE more fused pass: `redoFlag` is an INPUT here (as the first pass left
				  * it, with its one ghost cell filled: qk_FillBoundary_*_int) — a face that touches a flagged cell takes the first-order flux of
				  * U_old (donor cell + LLF, evaluated on demand; in stage 2 it replaces flux_rk2 of that face), a flagged cell takes the
				  * cell-centred velocity divergence in its P dV term (hydro_system.hpp:804-808); halfFlux keeps the uncorrected stage-1 fluxes;
				  * d_redo_count (zeroed by the caller) receives the cells that are STILL invalid; no flags are written.  Bit-identical to the
				  * reference-shaped operators (qk_hydro_ComputeFluxes(LLF) + qk_replaceFluxes + ...).  Requires K_visc == 0; in the carried-rhs
				  * form only stage 1 (the stage-1 pass ran with rk2_carry_rhs = 1; pass rk2_carry_rhs = 0 or 1 here: rhs1 is not touched) */
	int prim_out;		 /* stage 1 only.  1: U_out receives, per valid cell, the PRIMITIVES of the state the stage would have stored — (rho, v_x, v_y, v_z,
				  * P, E_int_aux) in components 0..5, HydroSystem::ConservedToPrimitive of it (src/hydro/hydro_system.hpp:138-196), passive scalars
				  * unchanged — which costs the final sweep one pressure (its limits already formed the velocities).  Ghost cells of such an array
				  * are filled by the same copies and the same reflect / extrapolate rules component by component (the momenta's parity is the
				  * velocities'); Dirichlet functors that write conserved values do not apply.  gamma law, reconstruct_eint = 0, no correction
				  * pass: a step in which either stage counts flagged cells is redone without the hand-off (U_old is untouched by both stages). */
	int prim_in;		 /* stage 2 only.  1: U_in holds what a stage 1 with prim_out stored: the pre-pass and the three sweeps read their primitives
				  * instead of converting the conserved state four times over (4.9 conversions per cell); same bytes, same bits. */
} qk_hydro_stage_args;

/* One RK stage of advanceHydroAtLevel (reference src/QuokkaSimulation.hpp:1099-1198 / 1202-1287) WITHOUT the
 * FOFC branch: computeHydroFluxes + Saxpy + ComputeRhsFromFluxes + AddInternalEnergyPdV + PredictStep +
 * EnforceLimits + SyncDualEnergy, fused per sweep direction.  If *d_redo_count > 0 afterwards the caller runs
 * the reference-shaped operators above for the FOFC correction (results are bit-identical where no flux is
 * replaced).
 * Launches per stage: the flattening pre-pass, the X, Y and Z sweeps — or, in the carried form (rk2_carry_rhs) of a 3-D level without passive scalars
 * and flux mask whose boxes are all a multiple of 64 cells wide, the pre-pass, ONE sweep that forms the x fluxes inside the y march, and the Z sweep:
 * same bits, 142 B per cell less HBM traffic.  Environment (read per call; A/B runs and tests): QK_FUSEX=0 keeps the four launches. */
int64_t qk_hydro_stage_scratch_bytes(qk_level *lev, const qk_hydro_traits *t);
int qk_hydro_stage_fused(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_hydro_stage_args *a);

/* ------------------------------------------------------------------ RadSystem<problem_t> (M1 closure, 1 .. QK_MAX_GROUPS photon groups) */
#define QK_MAX_GROUPS 8
/* RadSystem_Traits<P> (reference src/radiation/radiation_system.hpp:73-82, :201-223) + the device hooks a problem specialises
 * (ComputePlanckOpacity / ComputeEnergyMeanOpacity / ComputeFluxMeanOpacity, :1141-1154; DefineOpacityExponentsAndLowerValues, :1155-1167)
 * as closed parametrised sets. */
typedef struct qk_rad_traits {
	double c_light, c_hat, radiation_constant, Erad_floor;
	int beta_order;	   /* 0..3 (multigroup: 0 or 1, source_terms_multi_group.hpp:526) */
	int opacity_model; /* SINGLE GROUP (ngroups == 1, OpacityModel::single_group):
			    * 0: constants kappaP, kappaE, kappaF [cm^2 g^-1] (what RadhydroShell needs);
			    * 1: kappa = kappaX / rho, a constant absorption coefficient [cm^-1] (RadhydroShockCGS, test_radhydro_shock_cgs.cpp:78-86)
			    * 2: temperature power law  kappa = kappaX * max(pow(T / opacity_T_ref, opacity_T_exponent), opacity_pow_floor) / rho
			    *    (RadMarshakAsymptotic test_radiation_marshak_asymptotic.cpp:55-59: exponent -3; RadhydroPulseGrey: -3.5 with
			    *    different Planck and flux means; RadPulse: +3 with floor 1) */
	double kappaP, kappaE, kappaF;
	int pow_mode; /* 0: pow(T,4), pow(T,3) as the reference's std::pow; 1: repeated multiplication (bit-level tests; opacity exponents
		       * 3, -3 and -3.5 are then products / square roots as well) */
	int eddington_model; /* the ComputeEddingtonFactor hook: 0 Levermore closure (radiation_system.hpp:773-790), 1 chi = 1/3 */
	double opacity_T_ref, opacity_T_exponent, opacity_pow_floor; /* opacity_model 2 only (floor 0: none) */
	/* MULTIGROUP (Physics_Traits<P>::nGroups > 1; 0 is read as 1).  Group g owns state components radFirstIndex + 4 g .. + 4 g + 3. */
	int ngroups;
	int mg_opacity_model; /* OpacityModel (radiation_system.hpp:64-71): 1 piecewise_constant_opacity, 2 PPL_opacity_fixed_slope_spectrum,
			       * 3 PPL_opacity_full_spectrum */
	double energy_unit;			   /* RadSystem_Traits<P>::energy_unit: photon energy = energy_unit * boundary value */
	double rad_boundaries[QK_MAX_GROUPS + 1]; /* RadSystem_Traits<P>::radBoundaries (ngroups + 1 increasing edges) */
	/* DefineOpacityExponentsAndLowerValues(rad_boundaries, rho, T) as a closed set: for g = 0 .. ngroups
	 *   exponent[g]    = mg_kappa_exponent[g]
	 *   lower_value[g] = mg_kappa_lower[g] * pow(rho, mg_kappa_rho_exponent) * pow(T / mg_kappa_T_ref, mg_kappa_T_exponent)
	 * (either power with exponent 0 is skipped, exponent -1 on rho is a division).  Covers RadhydroShockMultigroup (577 / rho, exponent 0),
	 * RadTube / RadhydroPulseMGconst (constant), RadMarshakVaytet the_model 0 / 1 / 2 / 10 (kappa0 (nu_g / nu_pivot)^-2, exponent -2). */
	double mg_kappa_exponent[QK_MAX_GROUPS + 1], mg_kappa_lower[QK_MAX_GROUPS + 1];
	double mg_kappa_rho_exponent, mg_kappa_T_ref, mg_kappa_T_exponent;
	/* ISM_Traits<P>::enable_dust_gas_thermal_coupling_model (radiation_system.hpp:84-98): the dust temperature of Bate & Keto
	 * between gas and radiation (ComputeDustTemperatureBateKeto, :1420-1483) and its Jacobian (source_terms_single_group.hpp:165-175, :293-305).
	 * dust_gas_interaction_coeff = QuokkaSimulation::dustGasInteractionCoeff_ (QuokkaSimulation.hpp:127, default 2.5e-34 erg cm^3 s^-1 K^-3/2). */
	int enable_dust_gas_thermal_coupling_model;
	double dust_gas_interaction_coeff;
	/* the ComputeThermalRadiationSingleGroup / ...TempDerivativeSingleGroup hooks: 0 = a T^4 / 4 a T^3 (:471-479, :499-503); 1 = a T / a, the
	 * linearised emission of RadDust (src/problems/RadDust/test_rad_dust.cpp:86-97; accepted together with the dust model only) */
	int thermal_model;
	/* ISM_Traits<P>::gas_dust_coupling_threshold (radiation_system.hpp:89, default 1e-6; multigroup dust model only): when
	 * (c / c_hat) max(Gamma_gd) < threshold * E_gas the solve treats gas and dust as decoupled (radiation_dust_system.hpp:258-272) */
	double gas_dust_coupling_threshold;
	/* The ISM heating / cooling hooks of RadSystem<P> (radiation_system.hpp:344-353; defaults zero), as the closed set the reference's problems use
	 * (RadLineCooling, RadLineCoolingMG, RadMarshakDustPE); accepted together with the dust model only:
	 *   DefineNetCoolingRate(T, n)[g]               = cooling_linear_coeff[g] * T     (...TempDerivative = cooling_linear_coeff[g])
	 *   DefineCosmicRayHeatingRate(n)               = cr_heating_rate
	 *   DefinePhotoelectricHeatingE1Derivative(T, n) = pe_heating_E1_derivative, with ISM_Traits::enable_photoelectric_heating (multigroup: the
	 *   heating is proportional to the energy density of the LAST group; radiation_dust_system.hpp:578-933) */
	double cooling_linear_coeff[QK_MAX_GROUPS];
	double cr_heating_rate;
	int enable_photoelectric_heating;
	double pe_heating_E1_derivative;
} qk_rad_traits;
/* State layout: Physics_Indices (reference src/physics_info.hpp:20-47): comps 0..5 hydro, 6 + 4 g .. 9 + 4 g = (E_r, F_x, F_y, F_z) of group g.
 * The operators below act on all groups (primVar / flux arrays carry 4 * ngroups components, group-major like the state). */

/* ConservedToPrimitive(cons, primVar, indexRange = valid grown by nghost); primVar has 4 * ngroups comps   reference src/radiation/radiation_system.hpp:589-614 */
int qk_rad_ConservedToPrimitive(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_array4 *cons, qk_array4 *primVar, int nghost);
/* RadSystem::ComputeCellOpticalDepth<DIR> for every face, and what the use_wavespeed_correction branch of ComputeFluxes makes of it: eps[d] (face-centred
 * in d, no ghost cells, ngroups components) receives epsilon = min(1, 1 / tau_cell) on the faces whose i + j + k is even and 1 on the others, tau_cell the
 * harmonic mean of dl rho kappa of the face's two cells (kappa: ComputeFluxMeanOpacity for one group, ComputeBinCenterOpacity of
 * DefineOpacityExponentsAndLowerValues for several) with the gas temperature of quokka::EOS.  The transport entries below take these arrays as
 * `wavespeed_eps` (NULL = use_wavespeed_correction false, the reference's default, QuokkaSimulation.hpp:133).  Closed hook sets of qk_rad_traits; a
 * problem with compiled hooks instantiates the same kernel in its own translation unit (host/qk_problem_kernels.hpp).
 *                                                                       reference src/radiation/radiation_system.hpp:803-871, :1019-1022, :1098-1109 */
int qk_rad_ComputeWavespeedCorrection(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, int ndim, const qk_array4 *consVar,
				      const double dx[3], qk_array4 *const eps[3]);
/* ComputeFluxes<DIR>(x1Flux, x1FluxDiffusive (unused downstream: not produced), x1LeftState, x1RightState, x1FluxRange, consVar, dx,
 * use_wavespeed_correction); wavespeed_eps: NULL (false) or the array of direction `dir` from qk_rad_ComputeWavespeedCorrection(consVar)
 *                                                                                                reference src/radiation/radiation_system.hpp:985-1139 */
int qk_rad_ComputeFluxes(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int dir, qk_array4 *x1Flux, const qk_array4 *x1LeftState,
			 const qk_array4 *x1RightState, const qk_array4 *consVar, const qk_array4 *wavespeed_eps);
/* computeRadiationFluxes + fluxFunction<DIR> fused: cons -> (prim, reconstruction of `order` 1/2(MC)/3, HLL flux) in one kernel per direction
 *                                                                                                reference src/QuokkaSimulation.hpp:1884-1961 */
int qk_rad_computeRadiationFluxes(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int ndim, int reconstruction_order, const qk_array4 *consVar,
				  qk_array4 *const flux[3], const qk_array4 *const wavespeed_eps[3] /* NULL: no wavespeed correction */);
/* One transport stage without the face-flux round trip: computeRadiationFluxes(U_in) + PredictStep(U0 -> U_new) (stage 1) or
 * + AddFluxesRK2(U_new; U0, U1 = U_in) (stage 2, PD-ARS: the old-state fluxes have weight 0) with the flux divergence taken inside the flux
 * kernels (X, Y accumulate into `acc`: 4 components per cell, no ghost cells; Z finishes the update and repairs invalid states) — the state
 * equals the separate calls' in every bit.  U_new may be the arrays of U_in.  flux_out: NULL or three face-centred arrays that receive the
 * fluxes (flux registers).  3-D levels; with several photon groups (acc and the face arrays hold 4 components per group) one set of sweeps per group —
 * the groups are transported independently.  1-D / 2-D levels: QK_ERR_UNSUPPORTED (the separate calls serve them).
 *                                                    reference src/QuokkaSimulation.hpp:1726-1882, src/radiation/radiation_system.hpp:667-775 */
int qk_rad_stage_fused(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int reconstruction_order, int stage, const qk_array4 *U_in, const qk_array4 *U0,
		       qk_array4 *U_new, qk_array4 *acc, qk_array4 *const flux_out[3], double dt, const double dx[3],
		       const qk_array4 *const wavespeed_eps[3] /* NULL: no wavespeed correction; else of U_in */);
/* PredictStep(consVarOld, consVarNew, fluxArray, dt, dx, indexRange)                             reference src/radiation/radiation_system.hpp:667-710 */
int qk_rad_PredictStep(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int ndim, const qk_array4 *consVarOld, qk_array4 *consVarNew,
		       const qk_array4 *const fluxArray[3], double dt, const double dx[3]);
/* AddFluxesRK2(U_new, U0, U1, fluxArrayOld, fluxArray, dt, dx, indexRange)                       reference src/radiation/radiation_system.hpp:712-771 */
int qk_rad_AddFluxesRK2(qk_level *lev, qk_stream s, const qk_rad_traits *rt, int ndim, qk_array4 *U_new, const qk_array4 *U0, const qk_array4 *U1,
			const qk_array4 *const fluxArrayOld[3], const qk_array4 *const fluxArray[3], double dt, const double dx[3]);
/* AddSourceTermsSingleGroup / AddSourceTermsMultiGroup(consVar, radEnergySource, indexRange, dt, stage, dustGasCoeff, p_iteration_counter,
 * p_iteration_failure_counter)                                                                   reference src/radiation/source_terms_single_group.hpp:10-564
 *                                                                                                reference src/radiation/source_terms_multi_group.hpp:522-813
 * radEnergySource carries ngroups components.
 * d_iteration_counter: device int[4] (solves, Newton iterations, max Newton iterations, unused); d_failure_counter: device int[3]
 * (Newton failures, dust (unused), outer-iteration failures) — counted, never aborted, exactly as the reference. */
int qk_rad_AddSourceTermsSingleGroup(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *consVar,
				     const qk_array4 *radEnergySource, double dt, int stage, int *d_iteration_counter, int *d_failure_counter);
/* The same update, and the new radiation components (6 .. 9) of every valid cell stored into `mirror` as well: the swapRadiationState() that opens
 * the NEXT radiation substep (reference src/QuokkaSimulation.hpp:1783-1788: state_old <- state_new, radiation components) from the registers of
 * this kernel instead of a copy of 4 components through HBM.  Ghost cells of `mirror` are not written: advanceRadiationForwardEuler fills
 * all of them before it reads any (:1790-1795). */
int qk_rad_AddSourceTermsSingleGroupMirror(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *consVar,
					   const qk_array4 *radEnergySource, double dt, int stage, int *d_iteration_counter, int *d_failure_counter,
					   qk_array4 *mirror);
/* ngroups in {2, 3, 4, 5, 6, 8} (each is its own kernel instantiation: the per-group vectors live in registers); gas + radiation only (no dust /
 * photoelectric / cooling models, ISM_Traits defaults) */
int qk_rad_AddSourceTermsMultiGroup(qk_level *lev, qk_stream s, const qk_rad_traits *rt, const qk_hydro_traits *t, qk_array4 *consVar,
				    const qk_array4 *radEnergySource, double dt, int stage, int *d_iteration_counter, int *d_failure_counter);
/* device-side helper functions of the multigroup path, exposed for unit checks (host arrays in, host arrays out; one tiny launch each):
 * ComputePlanckEnergyFractions + ComputeThermalRadiationMultiGroup at temperature T (radiation_system.hpp:430-461, :483-497); kB = EOS_Traits::boltzmann_constant */
int qk_rad_mg_planck_fractions(qk_ctx *ctx, const qk_rad_traits *rt, double kB, int n, const double *T, double *fractions, double *Erad_g);

/* ------------------------------------------------------------------ level-0 ghost fill */
/* AMRSimulation::fillBoundaryConditions, level-0 branch                     reference src/simulation.hpp:1751-1776
 *   1. state.FillBoundary(geom.periodicity()) for neighbours on the same GPU (copy kernel)
 *      + pack / unpack of the strips exchanged with other GPUs (RCCL p2p is done by the caller);
 *   2. PhysBCFunct: amrex FilccCell (reflect_even/odd, foextrap) then the user functor
 *      (closed set: constant Dirichlet state per face, as HydroShocktube's setCustomBoundaryConditions). */
typedef struct qk_geometry {
	qk_box domain;
	int periodic[3];
	int ndim;
} qk_geometry;

typedef struct qk_bcrec {
	int lo[3];
	int hi[3];
} qk_bcrec; /* == amrex::BCRec per component */

/* user functor model (closed set): cells beyond face (dim,side) get the constant state `values[ncomp]` (all comps).
 * marshak != 0 (lower faces only): the Marshak half-range condition of the reference's RadMarshak family
 * (src/problems/RadMarshak/test_radiation_marshak.cpp:125-141): after the constant state, the normal radiation flux of the ghost cell
 * becomes  0.5 c E_inc - 0.5 (c E_0 + 2 F_0)  with E_inc = values[marshak_energy_comp] and (E_0, F_0) = components
 * (marshak_energy_comp, marshak_flux_comp) of the first valid cell inside the face (same transverse indices). */
#define QK_MAX_STATE_COMPS 48 /* 6 hydro + 3 scalars + 4 * QK_MAX_GROUPS radiation, rounded up */
typedef struct qk_dirichlet_face {
	int enabled;
	double values[QK_MAX_STATE_COMPS];
	int marshak;
	int marshak_energy_comp;
	int marshak_flux_comp;
	double marshak_c;
	/* RadTube's functor (src/problems/RadTube/test_radiation_tube.cpp:184-252): a component whose bit is set in interior_mask takes, instead of
	 * the constant, the value of the first valid cell inside the face (same transverse indices) — there the normal momentum and the normal
	 * radiation flux of every group.  kinetic_from_interior != 0: the total-energy component (4) becomes
	 * values[5] + 0.5 m^2 / values[0], m = the momentum of that cell normal to the face (values[5] = the constant internal energy). */
	uint64_t interior_mask;
	int kinetic_from_interior;
} qk_dirichlet_face;

typedef struct qk_ghost_plan qk_ghost_plan;
/* Build the exchange plan for `nghost` ghost cells.  `all_boxes`/`owner_rank` describe the whole level
 * (every rank builds the same plan); boxes with owner_rank == my_rank must be the level's boxes, in order. */
int qk_ghost_plan_create(qk_level *lev, qk_ghost_plan **plan, const qk_geometry *geom, int nghost, int ncomp, int n_all_boxes,
			 const qk_box *all_boxes, const int *owner_rank, int my_rank);
int qk_ghost_plan_destroy(qk_ghost_plan *plan);
/* remote traffic description: number of peers, and for peer k its rank and the number of doubles sent/received */
int qk_ghost_plan_num_peers(qk_ghost_plan *plan);
int qk_ghost_plan_peer(qk_ghost_plan *plan, int k, int *rank, int64_t *send_count, int64_t *recv_count);
/* plan introspection (host-side; what the kernels below execute).  kind 0: same-rank copies, 1: strips packed for peer k,
 * 2: strips unpacked from peer k, 3: ghost slabs beyond non-periodic domain faces.  Regions are in the DESTINATION index
 * space; source index = destination index - shift; offset = position (in elements) inside the peer buffer. */
int qk_ghost_plan_num_items(qk_ghost_plan *plan, int kind, int k);
int qk_ghost_plan_item(qk_ghost_plan *plan, int kind, int k, int idx, int *dst_box, int *src_box, int lo[3], int hi[3], int shift[3], int64_t *offset);
/* on-GPU copies (same-rank neighbours and periodic images) */
int qk_FillBoundary_local(qk_ghost_plan *plan, qk_stream s, qk_array4 *state);
/* iMultiFab flavour for redoFlag.FillBoundary (reference src/QuokkaSimulation.hpp:1157); plan built with ncomp = 1, nghost = 1 */
int qk_FillBoundary_local_int(qk_ghost_plan *plan, qk_stream s, qk_iarray4 *state);
/* pack the strips for peer k into `sendbuf` (device, send_count doubles) / unpack `recvbuf` */
int qk_FillBoundary_pack(qk_ghost_plan *plan, qk_stream s, int k, const qk_array4 *state, double *sendbuf);
int qk_FillBoundary_unpack(qk_ghost_plan *plan, qk_stream s, int k, qk_array4 *state, const double *recvbuf);
/* iMultiFab flavours (redoFlag strips between GPUs; counts of qk_ghost_plan_peer are elements, here 4-byte ints) */
int qk_FillBoundary_pack_int(qk_ghost_plan *plan, qk_stream s, int k, const qk_iarray4 *state, int *sendbuf);
int qk_FillBoundary_unpack_int(qk_ghost_plan *plan, qk_stream s, int k, qk_iarray4 *state, const int *recvbuf);
/* amrex::FabArray::SumBoundary with the same plan: every ghost value is ADDED to the valid cell it is a copy of.  Across ranks the wire
 * runs backwards: pack takes the strips this rank receives in FillBoundary (buffer of recv_count doubles), unpack adds a buffer of
 * send_count doubles to the valid cells this rank sends.  (Used for the reflux increments that land in ghost cells of a coarse box.) */
int qk_SumBoundary_local(qk_ghost_plan *plan, qk_stream s, qk_array4 *state);
int qk_SumBoundary_pack(qk_ghost_plan *plan, qk_stream s, int k, const qk_array4 *state, double *buf);
int qk_SumBoundary_unpack(qk_ghost_plan *plan, qk_stream s, int k, qk_array4 *state, const double *buf);
/* physical boundaries (after FillBoundary): bcs[ncomp]; dirichlet[dim][side] may be NULL */
int qk_FillPhysicalBoundary(qk_ghost_plan *plan, qk_stream s, qk_array4 *state, const qk_bcrec *bcs, const qk_dirichlet_face *dirichlet);
/* fillBoundaryConditions fills state.nComp() components (reference src/simulation.hpp:1755); the radiation substeps read only the radiation
 * components of the ghost cells: restrict the same-rank copies and the physical BCs that follow to [scomp, scomp + ncomp) (ncomp < 0: all) */
int qk_ghost_plan_set_components(qk_ghost_plan *plan, int scomp, int ncomp);
/* Overlap of the exchange with the update (north_star: "FillBoundary ... overlapped with interior-cell updates"; the
 * reference's FillBoundary, src/simulation.hpp:1755, is blocking).  A local box is "remote dependent" if any of its ghost
 * cells is filled from another rank.  The other boxes are complete after qk_FillBoundary_local + the LOCAL_ONLY subset
 * of the physical boundaries and can be advanced while the strips of the peers are on the wire; the REMOTE_DEPENDENT
 * subset runs after the unpack. */
#define QK_BOXES_ALL 0
#define QK_BOXES_LOCAL_ONLY 1
#define QK_BOXES_REMOTE_DEPENDENT 2
int qk_ghost_plan_box_is_remote(qk_ghost_plan *plan, int local_box); /* 1 / 0, < 0 on error */
/* move a box into / out of the late group by hand (load balancing between the two launches; single-GPU tests of the split) */
int qk_ghost_plan_set_box_remote(qk_ghost_plan *plan, int local_box, int flag);
int qk_FillPhysicalBoundary_subset(qk_ghost_plan *plan, qk_stream s, qk_array4 *state, const qk_bcrec *bcs, const qk_dirichlet_face *dirichlet,
				   int which);

/* ------------------------------------------------------------------ AMR level machinery: data-parallel pieces */
/* == amrex::Array4<char> (TagBox) */
typedef struct qk_carray4 {
	char *p;
	int64_t jstride, kstride, nstride;
	int begin[3];
	int end[3];
	int ncomp;
} qk_carray4;
enum { QK_TAG_CLEAR = 0, QK_TAG_BUF = 1, QK_TAG_SET = 2 }; /* amrex::TagBox::TagVal */
#define QK_TAGFIELD_PRESSURE (-1)			    /* HydroSystem<problem_t>::ComputePressure(state, i, j, k) */
/* QuokkaSimulation<problem_t>::ErrorEst of the gradient-threshold family
 *   reference src/problems/HydroBlast3D/test_hydro3d_blast.cpp:118-151 (field = pressure, P > P_min)
 *             src/problems/RadhydroShell/test_radhydro_shell.cpp:337-371 (field = density component, rho >= rho_min)
 * tags(i,j,k) = SET where  max_d max(|q(+e_d) - q|, |q - q(-e_d)|) / q > eta_threshold  and  q > q_min (>= if min_inclusive);
 * other cells are left untouched.  `state` needs one ghost cell.  field: QK_TAGFIELD_PRESSURE or a component index. */
int qk_tag_relative_gradient(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_array4 *state, qk_carray4 *tags, int field,
			     double eta_threshold, double q_min, int min_inclusive);
/* ErrorEst of HydroShocktube (reference src/problems/HydroShocktube/test_hydro_shocktube.cpp:146-170): centred difference of component
 * `comp` along `dir`:  del = (q(+1) - q(-1)) / (2 dx);  SET where sqrt(del^2) / q > eta_threshold and q >= q_min (> if !min_inclusive) */
int qk_tag_centered_gradient(qk_level *lev, qk_stream s, const qk_array4 *state, qk_carray4 *tags, int comp, int dir, double dx, double eta_threshold,
			     double q_min, int min_inclusive);
/* QuokkaSimulation::PreInterpState / PostInterpState(mf, scomp, ncomp)     reference src/QuokkaSimulation.hpp:804-841
 * around the coarse-to-fine interpolation: E <- (E - |p|^2/(2 rho)) / rho on every valid cell of `mf`, and back (E <- rho e + KE). */
int qk_PreInterpState(qk_level *lev, qk_stream s, qk_array4 *mf);
int qk_PostInterpState(qk_level *lev, qk_stream s, qk_array4 *mf);
/* AMRSimulation::AverageDownTo -> amrex::average_down(fine, crse, ...)     reference src/simulation.hpp:1949-1964
 * crse = (1 / (rx ry rz)) * sum of the fine cells under it, summed with the x index fastest (amrex_avgdown); only coarse cells
 * covered by a fine box are written.  The plan pairs every fine box with the coarse boxes it overlaps (same rank). */
typedef struct qk_avgdown_plan qk_avgdown_plan;
int qk_avgdown_plan_create(qk_level *crse, qk_level *fine, const int ratio[3], qk_avgdown_plan **plan);
int qk_avgdown_plan_destroy(qk_avgdown_plan *plan);
int qk_avgdown_plan_num_items(qk_avgdown_plan *plan);
int qk_average_down(qk_avgdown_plan *plan, qk_stream s, const qk_array4 *fine, qk_array4 *crse, int scomp, int ncomp);

/* Coarse -> fine part of AMRSimulation::FillPatchWithData / amrex::FillPatchTwoLevels       reference src/simulation.hpp:1789-1858
 * Fine cells that no fine box covers (ghost cells inside the domain or beyond a periodic face; whole_fab != 0: every cell of
 * the grown fine boxes, used when a level is (re)made) are interpolated from  w_old * crse_old + w_new * crse_new
 * (FillPatch time interpolation; pass the same table twice and w_new = 0 for a single time level):
 *   method 1 = amrex::mf_linear_slope_minmax_interp, 0 = amrex::mf_pc_interp (amrInterpMethod_, src/simulation.hpp:1389-1401);
 *   energy_hooks != 0 wraps the interpolation in QuokkaSimulation::PreInterpState / PostInterpState (src/QuokkaSimulation.hpp:804-841).
 * The coarse arrays need their ghost cells filled (stencil: one coarse cell around the parent cell).  AMReX's interpolater is
 * restated from its documentation: parity with the reference is unpinned for this entry point. */
typedef struct qk_interp_plan qk_interp_plan;
/* all_fine (n_all_fine boxes, may be NULL = the boxes of `fine`): the fine boxes of ALL ranks — ghost cells under a remote fine box
 * are filled by the fine-fine exchange, not by interpolation */
int qk_interp_plan_create(qk_level *crse, qk_level *fine, const qk_geometry *fine_geom, int nghost, const int ratio[3], int whole_fab, int n_all_fine,
			  const qk_box *all_fine, qk_interp_plan **plan);
int qk_interp_plan_destroy(qk_interp_plan *plan);
int qk_interp_plan_num_items(qk_interp_plan *plan);
int qk_interp_plan_item(qk_interp_plan *plan, int idx, int *fine_box, int *crse_box, int lo[3], int hi[3]);
int qk_InterpFromCoarse(qk_interp_plan *plan, qk_stream s, qk_array4 *fine, const qk_array4 *crse_old, const qk_array4 *crse_new, double w_old,
			double w_new, int ncomp, int method, int energy_hooks);

/* amrex::YAFluxRegister between a coarse level and the next finer one, as driven by AMRSimulation::incrementFluxRegisters
 * (reference src/simulation.hpp:1345-1387: CrseAdd / FineAdd with the level's fluxes, cell size and dt) and
 * timeStepWithSubcycling (:1308: Reflux(state_new_cc_[lev])).  Register cells: coarse cells just outside a fine box, not under
 * another fine box, inside the (periodic) domain.  See qk_amr_fluxreg.hip for the accumulated expression.  Parity unpinned
 * (AMReX's kernel is restated). */
typedef struct qk_fluxreg qk_fluxreg;
/* all_fine as for qk_interp_plan_create.  reg_nghost > 0: a register cell owned by another rank is kept in a ghost cell (within
 * reg_nghost) of the local coarse box it adjoins IN THE DIRECTION OF ITS FACE — the box whose flux array holds the face CrseAdd reads (never a box that
 * merely has the cell in a corner of its ghost ring); Reflux must then target a zeroed increment array with that many ghost cells, to be folded with
 * qk_SumBoundary_* and added to the state (quokka_amd/amr_simulation.py).  reg_nghost = 0: single rank, Reflux straight into the state. */
int qk_fluxreg_create(qk_level *crse, qk_level *fine, const qk_geometry *crse_geom, const int ratio[3], int ncomp, int n_all_fine, const qk_box *all_fine,
		      int reg_nghost, qk_fluxreg **fr);
int qk_fluxreg_destroy(qk_fluxreg *fr);
int qk_fluxreg_num_items(qk_fluxreg *fr);
int qk_fluxreg_item(qk_fluxreg *fr, int idx, int *dir, int *side, int *fine_box, int *crse_box, int lo[3], int hi[3], int shift[3]);
int qk_fluxreg_reset(qk_fluxreg *fr, qk_stream s);
/* fr_as_fine->getFineData() saved before the retry loop of the fine level and copied back at every retry
 * (reference src/QuokkaSimulation.hpp:894-900, :926-928); the copy lives inside the register object */
int qk_fluxreg_save(qk_fluxreg *fr, qk_stream s);
int qk_fluxreg_restore(qk_fluxreg *fr, qk_stream s);
int qk_fluxreg_CrseAdd(qk_fluxreg *fr, qk_stream s, const qk_array4 *const flux[3], const double dx[3], double dt);
int qk_fluxreg_FineAdd(qk_fluxreg *fr, qk_stream s, const qk_array4 *const flux[3], const double dx_fine[3], double dt);
int qk_fluxreg_Reflux(qk_fluxreg *fr, qk_stream s, qk_array4 *crse_state);
/* A register that covers a component range of the state (the radiation block: the reference's expandFluxArrays, QuokkaSimulation.hpp:1758, places
 * the radiation fluxes at nstartHyperbolic_ of a full-width flux array): Reflux adds register component n to state component comp0 + n. */
int qk_fluxreg_set_state_component(qk_fluxreg *fr, int comp0);

/* The COARSE side of a register whose fine level is distributed independently of the coarse one (AMReX hands every level its own
 * DistributionMapping, reference src/simulation.hpp:1421-1500; amrex::YAFluxRegister keeps m_crse_data on the coarse level's distribution and
 * m_cfpatch on the fine level's): `fine` is a level object holding the fine boxes of ALL ranks (box metadata only), register cells that no LOCAL
 * coarse box holds are skipped — their owner builds them.  Takes CrseAdd and Reflux (straight into the local coarse state).  The FINE side is an
 * ordinary qk_fluxreg_create(crse = the coarsened local fine boxes ("shadow" level), fine = the local fine boxes, all_fine, reg_nghost = 1):
 * FineAdd, then Reflux into a zeroed shadow array whose one-cell ghost ring travels to the coarse owners with qk_ParallelCopy_* (add = 1). */
int qk_fluxreg_create_crse_part(qk_level *crse, qk_level *all_fine_level, const qk_geometry *crse_geom, const int ratio[3], int ncomp, qk_fluxreg **fr);

/* amrex::FabArray::ParallelCopy / ParallelAdd between two box layouts with independent owners (qk_amr_pcopy.hip): every cell of the destination
 * boxes grown by dst_nghost (minus dst_holes[b], if given: a box per destination box whose cells are NOT wanted) takes the value of the source
 * cell with the same index — or its periodic image — found in a source box grown by src_nghost (src_ring_only != 0: in its ghost ring alone).
 * add = 0: copy (the source pieces must not overlap: src_nghost = 0); add != 0: accumulate (atomic: pieces of several source boxes may meet in
 * one cell).  The box lists and owners describe ALL ranks and are the same everywhere; the tables passed to the data calls hold this rank's boxes
 * of each list, in list order.  Peer buffers hold `ncomp` values per cell (send_count / recv_count of qk_pcopy_plan_peer are values); the calls
 * move components [scomp_src, scomp_src + ncomp) to [scomp_dst, ...).  Same wire protocol as the ghost plan: pack -> one send / recv pair per
 * peer -> local -> unpack.  Reference: the FillPatchTwoLevels / average_down / YAFluxRegister::Reflux / RemakeLevel data motion of
 * src/simulation.hpp:1789-1858, :1949-1964, :1308, :1672-1685 when levels have their own DistributionMapping (:1421-1500, :1657-1702).
 * QK_GHOST_LOOPBACK=1 (environment, read at plan creation as by qk_ghost_plan_create): same-rank pairs become regions of a peer whose rank is my_rank —
 * the pack -> send / recv -> unpack ordering of a plan on the production transport with ONE GPU (tests/test_rccl_loopback_gpu.py). */
typedef struct qk_pcopy_plan qk_pcopy_plan;
int qk_pcopy_plan_create(qk_ctx *ctx, const qk_geometry *geom, int n_src, const qk_box *src_boxes, const int *src_owner, int src_nghost, int src_ring_only,
			 int n_dst, const qk_box *dst_boxes, const int *dst_owner, int dst_nghost, const qk_box *dst_holes, int ncomp, int my_rank,
			 qk_pcopy_plan **plan);
int qk_pcopy_plan_destroy(qk_pcopy_plan *plan);
int qk_pcopy_plan_num_peers(qk_pcopy_plan *plan);
int qk_pcopy_plan_peer(qk_pcopy_plan *plan, int k, int *rank, int64_t *send_count, int64_t *recv_count);
/* introspection: kind 0 same-rank items, 1 packed for peer k, 2 unpacked from peer k (regions in the destination index space) */
int qk_pcopy_plan_num_items(qk_pcopy_plan *plan, int kind, int k);
int qk_pcopy_plan_item(qk_pcopy_plan *plan, int kind, int k, int idx, int *dst_box, int *src_box, int lo[3], int hi[3], int shift[3], int64_t *offset);
int qk_ParallelCopy_local(qk_pcopy_plan *plan, qk_stream s, const qk_array4 *src, qk_array4 *dst, int scomp_src, int scomp_dst, int add);
int qk_ParallelCopy_pack(qk_pcopy_plan *plan, qk_stream s, int k, const qk_array4 *src, int scomp_src, double *sendbuf);
int qk_ParallelCopy_unpack(qk_pcopy_plan *plan, qk_stream s, int k, qk_array4 *dst, int scomp_dst, const double *recvbuf, int add);

/* copy the region [lo, hi] (same index space) between two arrays given by HOST copies of their descriptors (device data):
 * the old-level data a remade level keeps (RemakeLevel's FillPatch copies fine data where it exists, reference src/simulation.hpp:1672-1685) */
int qk_copy_box(qk_ctx *ctx, qk_stream s, const qk_array4 *src, const qk_array4 *dst, const int lo[3], const int hi[3], int scomp, int dcomp, int ncomp);
/* Grid generation (amrex::AmrCore::MakeNewGrids is Berger-Rigoutsos clustering and not vendored; this is a simpler rule with the same
 * inputs amr.n_error_buf / amr.blocking_factor / amr.max_grid_size — grids differ from AMReX's, parity unpinned):
 *   qk_amr_tile_flags    tags (TagBox::SET) buffered by n_error_buf cells -> one int per tile of `tile` cells on a side (level index
 *                        space of `tags`, domain starting at 0), copied to the HOST array tile_flags_host[ntz][nty][ntx]; synchronises s
 *   qk_amr_cluster_tiles host: flagged tiles (tile = blocking_factor FINE cells) -> fine boxes, greedy merge x, y, z up to max_grid_size */
int qk_amr_tile_flags(qk_level *lev, qk_stream s, const qk_carray4 *tags, const qk_box *domain, int n_error_buf, int tile, int *tile_flags_host);
/* the same with the buffer carried through periodic faces of the domain (periodic[d] != 0), as AMReX maps buffered tags back into a periodic
 * domain (TagBoxArray::mapPeriodicRemoveDuplicates): a feature about to leave through a periodic face is refined where it re-enters */
int qk_amr_tile_flags_periodic(qk_level *lev, qk_stream s, const qk_carray4 *tags, const qk_box *domain, const int periodic[3], int n_error_buf, int tile,
			       int *tile_flags_host);
int qk_amr_cluster_tiles(const int *tiles, const int ntiles[3], int ndim, int blocking_factor, int max_grid_size, int parent_align, qk_box *boxes,
			 int max_boxes);
/* host: the same flagged tiles -> fine boxes by Berger-Rigoutsos point clustering with efficiency grid_eff (amr.grid_eff), then
 * BoxList::simplify and BoxList::maxSize(max_grid_size) — the steps of amrex::AmrMesh::MakeNewGrids after the tags have been buffered and
 * coarsened by blocking_factor / ref_ratio (reference tests/blast_amr_maxlev2.in:16-21; AMReX not vendored: restated, parity unpinned).
 * `allowed` (NULL: everywhere): tiles where refinement may go (the proper-nesting domain); a cluster box holding a forbidden tile is bisected
 * until every piece is allowed (ClusterList::intersect).  Returns the number of boxes (fine index space, edges multiples of blocking_factor) or a negative error code. */
int qk_amr_cluster_berger_rigoutsos(const int *tiles, const int *allowed, const int ntiles[3], int ndim, int blocking_factor, int max_grid_size, double grid_eff,
				    qk_box *boxes, int max_boxes);

/* ------------------------------------------------------------------ optically-thin cooling from Cloudy tables (Strang-split source)
 * The five arrays of quokka::TabulatedCooling::cloudy_tables (reference src/cooling/TabulatedCooling.hpp:54-73): log10 n_H (n_nH values,
 * uniformly spaced), log10 T (n_Tgas values), and three n_nH x n_Tgas tables with the n_H index running fastest — FastMath::log10 of the cooling
 * and heating rates in units of (1.67e-24 g)^2, and the dimensionless mean molecular weight —; the temperature and mean-molecular-weight ranges. */
typedef struct qk_cloudy_tables {
	const double *log_nH;
	const double *log_Tgas;
	const double *cooling;
	const double *heating;
	const double *mean_mol_weight;
	int n_nH, n_Tgas;
	double T_min, T_max;
	double mmw_min, mmw_max;
} qk_cloudy_tables;
/* readCloudyData(hdf5_file, cloudyTables) (reference src/cooling/TabulatedCooling.cpp:9-31 with initialize_cloudy_data,
 * src/cooling/CloudyDataReader.cpp:24-200): reads /Cooling (attributes Rank, Dimension), /Heating, /MMW, /Parameter1, /Temperature of a
 * cloudy_cooling_tools file — with the library's own reader of the HDF5 file format, libhdf5 is not needed — into HOST arrays owned by the library
 * (qk_cloudy_tables_free releases them).  The caller copies them to device memory and passes a struct of device pointers to the two kernels. */
int qk_cloudy_tables_read(qk_ctx *ctx, const char *path, qk_cloudy_tables *host_tables);
int qk_cloudy_tables_free(qk_cloudy_tables *host_tables);
/* quokka::TabulatedCooling::computeCooling<problem_t>(mf, dt, cloudyTables, T_floor) (reference src/cooling/TabulatedCooling.hpp:258-317), which
 * addStrangSplitSourcesWithBuiltin calls with dt = dt_lev / 2 before and after the hydro update (src/QuokkaSimulation.hpp:520-547,1048,1318):
 * dE_int/dt = (rho X)^2 (Gamma - Lambda)(n_H, T(E_int)) integrated over dt in every valid cell with the adaptive Heun integrator of
 * src/math/ODEIntegrate.hpp (rtol 1e-4, abstol 0.01 E(T_floor)); gas energy and auxiliary internal energy take the change.
 * d_counters (device, cleared by the caller): [0] the largest number of substeps any cell took, [1] their sum.  [0] == 2000
 * (maxStepsODEIntegrate) means the integration failed in some cell: the reference retries the hydro step with a smaller dt. */
int qk_cooling_tabulated(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *state, const qk_cloudy_tables *device_tables, double dt, double T_floor,
			 long long *d_counters);
/* The per-cell functions of src/cooling/TabulatedCooling.hpp a problem evaluates in its own kernels, over n (rho, value) pairs in device memory:
 * ComputeTgasFromEgas (:117-174; value = E_int), ComputeEgasFromTgas (:101-115; value = T), ComputeMMW (:206-220; value = E_int),
 * ComputeCoolingLength (:176-204; value = E_int), cloudy_cooling_function (:82-99; value = T). */
enum { QK_COOLING_TGAS_FROM_EGAS = 0, QK_COOLING_EGAS_FROM_TGAS = 1, QK_COOLING_MMW = 2, QK_COOLING_LENGTH = 3, QK_COOLING_NET_HEATING = 4 };
int qk_cooling_evaluate(qk_ctx *ctx, qk_stream s, const qk_cloudy_tables *device_tables, double gamma, int what, int64_t n, const double *d_rho, const double *d_value,
			double *d_out);

#ifdef __cplusplus
}
#endif
#endif /* QUOKKA_AMD_H_ */
