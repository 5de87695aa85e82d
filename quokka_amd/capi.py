"""ctypes binding of libquokka_amd.so (the C-ABI declared in include/quokka_amd.h).

This module is plumbing only: it loads the HIP library and declares prototypes.  There is NO CPU
fallback — if the library (or a GPU) is missing, importing the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QK_LIB_PATH", os.path.join(_HERE, "lib", "libquokka_amd.so"))  # (override: A/B runs of two builds on one box)

QK_OK, QK_ERR_INVALID, QK_ERR_HIP, QK_ERR_UNSUPPORTED, QK_ERR_STATE = 0, -1, -2, -3, -4
ERR_UNSUPPORTED = QK_ERR_UNSUPPORTED
HOOK_COMPILED = 100  # QK_HOOK_COMPILED: a hook that is the problem's own compiled device function (the library entry points refuse to evaluate it)
DIR_X1, DIR_X2, DIR_X3 = 0, 1, 2
RIEMANN_HLLC, RIEMANN_LLF, RIEMANN_HLLD = 0, 1, 2
LIMITER_MINMOD, LIMITER_MC = 0, 1
BC_REFLECT_ODD, BC_INT_DIR, BC_REFLECT_EVEN, BC_FOEXTRAP, BC_EXT_DIR = -1, 0, 1, 2, 3

K_B = 1.380649e-16
M_U = 1.6605390666e-24


class Array4(C.Structure):
    """== qk_array4 == amrex::Array4<double>"""
    _fields_ = [("p", C.c_void_p), ("jstride", C.c_int64), ("kstride", C.c_int64), ("nstride", C.c_int64),
                ("begin", C.c_int * 3), ("end", C.c_int * 3), ("ncomp", C.c_int)]


class Box(C.Structure):
    _fields_ = [("lo", C.c_int * 3), ("hi", C.c_int * 3)]


class HydroTraits(C.Structure):
    _fields_ = [("gamma", C.c_double), ("cs_isothermal", C.c_double), ("mean_molecular_weight", C.c_double),
                ("boltzmann_constant", C.c_double), ("reconstruct_eint", C.c_int), ("nscalars", C.c_int),
                ("nmscalars", C.c_int), ("ndim", C.c_int), ("eos_temperature_model", C.c_int), ("eos_alpha", C.c_double)]


class Geometry(C.Structure):
    _fields_ = [("domain", Box), ("periodic", C.c_int * 3), ("ndim", C.c_int)]


class BCRec(C.Structure):
    _fields_ = [("lo", C.c_int * 3), ("hi", C.c_int * 3)]


MAX_GROUPS = 8        # QK_MAX_GROUPS
MAX_STATE_COMPS = 48  # QK_MAX_STATE_COMPS


class DirichletFace(C.Structure):
    _fields_ = [("enabled", C.c_int), ("values", C.c_double * MAX_STATE_COMPS),
                ("marshak", C.c_int), ("marshak_energy_comp", C.c_int), ("marshak_flux_comp", C.c_int), ("marshak_c", C.c_double),
                ("interior_mask", C.c_uint64), ("kinetic_from_interior", C.c_int)]


class RadTraits(C.Structure):
    _fields_ = [("c_light", C.c_double), ("c_hat", C.c_double), ("radiation_constant", C.c_double), ("Erad_floor", C.c_double),
                ("beta_order", C.c_int), ("opacity_model", C.c_int), ("kappaP", C.c_double), ("kappaE", C.c_double), ("kappaF", C.c_double),
                ("pow_mode", C.c_int), ("eddington_model", C.c_int),
                ("opacity_T_ref", C.c_double), ("opacity_T_exponent", C.c_double), ("opacity_pow_floor", C.c_double),
                # multigroup (include/quokka_amd.h): Physics_Traits::nGroups, OpacityModel, energy_unit, radBoundaries and the
                # DefineOpacityExponentsAndLowerValues closed set
                ("ngroups", C.c_int), ("mg_opacity_model", C.c_int), ("energy_unit", C.c_double), ("rad_boundaries", C.c_double * (MAX_GROUPS + 1)),
                ("mg_kappa_exponent", C.c_double * (MAX_GROUPS + 1)), ("mg_kappa_lower", C.c_double * (MAX_GROUPS + 1)),
                ("mg_kappa_rho_exponent", C.c_double), ("mg_kappa_T_ref", C.c_double), ("mg_kappa_T_exponent", C.c_double),
                # ISM_Traits::enable_dust_gas_thermal_coupling_model, QuokkaSimulation::dustGasInteractionCoeff_, thermal-emission hook
                ("enable_dust_gas_thermal_coupling_model", C.c_int), ("dust_gas_interaction_coeff", C.c_double), ("thermal_model", C.c_int),
                ("gas_dust_coupling_threshold", C.c_double),
                # ISM hooks (closed set): line cooling linear in T per group, cosmic-ray heating, photoelectric heating by the last group
                ("cooling_linear_coeff", C.c_double * MAX_GROUPS), ("cr_heating_rate", C.c_double), ("enable_photoelectric_heating", C.c_int),
                ("pe_heating_E1_derivative", C.c_double)]

    def set_groups(self, boundaries, energy_unit, opacity_model, kappa_exponent, kappa_lower, rho_exponent=0.0, T_ref=0.0, T_exponent=0.0):
        """RadSystem_Traits<P>::radBoundaries / energy_unit / opacity_model + the DefineOpacityExponentsAndLowerValues hook (closed set)"""
        ng = len(boundaries) - 1
        assert 2 <= ng <= MAX_GROUPS and len(kappa_exponent) == len(kappa_lower) == ng + 1
        self.ngroups, self.mg_opacity_model, self.energy_unit = ng, int(opacity_model), float(energy_unit)
        for g in range(ng + 1):
            self.rad_boundaries[g] = float(boundaries[g])
            self.mg_kappa_exponent[g] = float(kappa_exponent[g])
            self.mg_kappa_lower[g] = float(kappa_lower[g])
        self.mg_kappa_rho_exponent, self.mg_kappa_T_ref, self.mg_kappa_T_exponent = float(rho_exponent), float(T_ref), float(T_exponent)
        return self


class CloudyTables(C.Structure):
    """qk_cloudy_tables: five arrays (host pointers from qk_cloudy_tables_read, or device pointers for the kernels) + ranges"""
    _fields_ = [("log_nH", C.c_void_p), ("log_Tgas", C.c_void_p), ("cooling", C.c_void_p), ("heating", C.c_void_p), ("mean_mol_weight", C.c_void_p),
                ("n_nH", C.c_int), ("n_Tgas", C.c_int), ("T_min", C.c_double), ("T_max", C.c_double), ("mmw_min", C.c_double), ("mmw_max", C.c_double)]


class StageArgs(C.Structure):
    _fields_ = [("U_in", C.c_void_p), ("U_old", C.c_void_p), ("U_out", C.c_void_p),
                ("halfFlux", C.c_void_p * 3), ("halfVel", C.c_void_p * 3),
                ("redoFlag", C.c_void_p), ("d_redo_count", C.c_void_p), ("d_error_flag", C.c_void_p), ("d_max_signal", C.c_void_p),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_int64),
                ("dx", C.c_double * 3), ("dt", C.c_double), ("stage", C.c_int), ("reconstruction_order", C.c_int),
                ("densityFloor", C.c_double), ("tempFloor", C.c_double), ("use_dual_energy", C.c_int), ("K_visc", C.c_double),
                ("store_flux_rk2", C.c_int), ("fluxRk2", C.c_void_p * 3), ("rk2_carry_rhs", C.c_int), ("rhs1", C.c_void_p), ("flux_mask", C.c_void_p), ("fofc_pass", C.c_int),
                ("prim_out", C.c_int), ("prim_in", C.c_int)]


def traits(gamma=1.4, reconstruct_eint=True, ndim=3, mean_molecular_weight=M_U, boltzmann_constant=K_B,
           cs_isothermal=float("nan"), nscalars=0, nmscalars=0, eos_temperature_model=0, eos_alpha=0.0) -> HydroTraits:
    return HydroTraits(gamma, cs_isothermal, mean_molecular_weight, boltzmann_constant, int(reconstruct_eint), nscalars, nmscalars, ndim,
                       int(eos_temperature_model), float(eos_alpha))


class QkError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load libquokka_amd.so; raise loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QkError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950). The hot path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    P = C.POINTER
    L.qk_version.restype = C.c_char_p
    L.qk_last_error.restype = C.c_char_p
    L.qk_last_error.argtypes = [vp]
    L.qk_ctx_create.argtypes = [P(vp), ci]
    L.qk_ctx_destroy.argtypes = [vp]
    L.qk_level_create.argtypes = [vp, P(vp), ci, ci, P(Box)]
    L.qk_level_destroy.argtypes = [vp]
    L.qk_upload_array4_table.argtypes = [vp, ci, P(Array4), P(vp)]
    L.qk_upload_iarray4_table.argtypes = [vp, ci, P(Array4), P(vp)]
    L.qk_profile_enable.argtypes = [vp, ci]
    L.qk_profile_only.argtypes = [vp, C.c_char_p]
    L.qk_clear_bytes.argtypes = [vp, vp, vp, C.c_int64]
    L.qk_profile_reset.argtypes = [vp]
    L.qk_cloudy_tables_read.argtypes = [vp, C.c_char_p, P(CloudyTables)]
    L.qk_cloudy_tables_free.argtypes = [P(CloudyTables)]
    L.qk_cooling_tabulated.argtypes = [vp, vp, P(HydroTraits), vp, P(CloudyTables), cd, cd, vp]
    L.qk_cooling_evaluate.argtypes = [vp, vp, P(CloudyTables), cd, ci, C.c_int64, vp, vp, vp]
    L.qk_profile_num_kernels.argtypes = [vp]
    L.qk_profile_get.argtypes = [vp, ci, P(C.c_char_p), P(C.c_long), P(cd)]
    T = P(HydroTraits)
    L.qk_ReconstructStatesConstant.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci]
    L.qk_ReconstructStatesPLM.argtypes = [vp, vp, ci, ci, vp, vp, vp, ci, ci]
    L.qk_ReconstructStatesPPM.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci, ci, ci]
    L.qk_hydro_ConservedToPrimitive.argtypes = [vp, vp, T, vp, vp, ci]
    L.qk_hydro_ComputeFlatteningCoefficients.argtypes = [vp, vp, T, ci, vp, vp, ci]
    L.qk_hydro_FlattenShocks.argtypes = [vp, vp, T, ci, vp, vp, vp, vp, vp, vp, ci, ci]
    L.qk_hydro_ComputeFluxes.argtypes = [vp, vp, T, ci, ci, vp, vp, vp, vp, vp, cd]
    L.qk_hydro_ComputeRhsFromFluxes.argtypes = [vp, vp, T, vp, P(vp), P(cd), ci]
    L.qk_hydro_AddInternalEnergyPdV.argtypes = [vp, vp, T, vp, vp, P(cd), P(vp), vp]
    L.qk_hydro_PredictStep.argtypes = [vp, vp, T, vp, vp, vp, cd, ci, vp, vp]
    L.qk_hydro_EnforceLimits.argtypes = [vp, vp, T, cd, cd, vp]
    L.qk_hydro_SyncDualEnergy.argtypes = [vp, vp, T, vp, vp]
    L.qk_hydro_ComputeMaxSignalSpeed.argtypes = [vp, vp, T, vp, vp]
    L.qk_hydro_maxSignalSpeedLocal.argtypes = [vp, vp, T, ci, vp, vp]
    L.qk_replaceFluxes.argtypes = [vp, vp, ci, vp, vp, vp, ci]
    L.qk_Saxpy.argtypes = [vp, vp, ci, vp, cd, vp, ci]
    R = P(RadTraits)
    L.qk_rad_ConservedToPrimitive.argtypes = [vp, vp, R, vp, vp, ci]
    L.qk_rad_ComputeFluxes.argtypes = [vp, vp, R, ci, vp, vp, vp, vp, vp]
    L.qk_rad_computeRadiationFluxes.argtypes = [vp, vp, R, ci, ci, vp, P(vp), P(vp)]
    L.qk_rad_ComputeWavespeedCorrection.argtypes = [vp, vp, R, T, ci, vp, P(cd), P(vp)]
    L.qk_rad_PredictStep.argtypes = [vp, vp, R, ci, vp, vp, P(vp), cd, P(cd)]
    L.qk_rad_AddFluxesRK2.argtypes = [vp, vp, R, ci, vp, vp, vp, P(vp), P(vp), cd, P(cd)]
    L.qk_hydro_FixupState.argtypes = [vp, vp, T, cd, cd, ci, vp, vp, vp]
    L.qk_rad_stage_fused.argtypes = [vp, vp, R, ci, ci, vp, vp, vp, vp, P(vp), cd, P(cd), P(vp)]
    L.qk_rad_AddSourceTermsSingleGroup.argtypes = [vp, vp, R, T, vp, vp, cd, ci, vp, vp]
    L.qk_rad_AddSourceTermsSingleGroupMirror.argtypes = [vp, vp, R, T, vp, vp, cd, ci, vp, vp, vp]
    L.qk_rad_AddSourceTermsMultiGroup.argtypes = [vp, vp, R, T, vp, vp, cd, ci, vp, vp]
    L.qk_rad_mg_planck_fractions.argtypes = [vp, R, cd, ci, P(cd), P(cd), P(cd)]
    if hasattr(L, "qk_hydro_stage_fused"):
        L.qk_hydro_stage_scratch_bytes.argtypes = [vp, T]
        L.qk_hydro_stage_scratch_bytes.restype = C.c_int64
        L.qk_hydro_stage_fused.argtypes = [vp, vp, T, P(StageArgs)]
    if hasattr(L, "qk_ghost_plan_create"):
        L.qk_ghost_plan_create.argtypes = [vp, P(vp), P(Geometry), ci, ci, ci, P(Box), P(ci), ci]
        L.qk_ghost_plan_destroy.argtypes = [vp]
        L.qk_ghost_plan_num_peers.argtypes = [vp]
        L.qk_ghost_plan_peer.argtypes = [vp, ci, P(ci), P(C.c_int64), P(C.c_int64)]
        L.qk_ghost_plan_num_items.argtypes = [vp, ci, ci]
        L.qk_ghost_plan_item.argtypes = [vp, ci, ci, ci, P(ci), P(ci), ci * 3, ci * 3, ci * 3, P(C.c_int64)]
        L.qk_FillBoundary_local.argtypes = [vp, vp, vp]
        L.qk_FillBoundary_local_int.argtypes = [vp, vp, vp]
        L.qk_FillBoundary_pack.argtypes = [vp, vp, ci, vp, vp]
        L.qk_FillBoundary_unpack.argtypes = [vp, vp, ci, vp, vp]
        L.qk_SumBoundary_local.argtypes = [vp, vp, vp]
        L.qk_SumBoundary_pack.argtypes = [vp, vp, ci, vp, vp]
        L.qk_SumBoundary_unpack.argtypes = [vp, vp, ci, vp, vp]
        L.qk_FillBoundary_pack_int.argtypes = [vp, vp, ci, vp, vp]
        L.qk_FillBoundary_unpack_int.argtypes = [vp, vp, ci, vp, vp]
        L.qk_FillPhysicalBoundary.argtypes = [vp, vp, vp, P(BCRec), P(DirichletFace)]
        L.qk_FillPhysicalBoundary_subset.argtypes = [vp, vp, vp, P(BCRec), P(DirichletFace), C.c_int]
        L.qk_ghost_plan_set_components.argtypes = [vp, C.c_int, C.c_int]
        L.qk_ghost_plan_box_is_remote.argtypes = [vp, C.c_int]
        L.qk_ghost_plan_set_box_remote.argtypes = [vp, C.c_int, C.c_int]
    if hasattr(L, "qk_tag_relative_gradient"):
        L.qk_tag_relative_gradient.argtypes = [vp, vp, T, vp, vp, ci, C.c_double, C.c_double, ci]
        L.qk_tag_centered_gradient.argtypes = [vp, vp, vp, vp, ci, ci, C.c_double, C.c_double, C.c_double, ci]
        L.qk_avgdown_plan_create.argtypes = [vp, vp, ci * 3, P(vp)]
        L.qk_avgdown_plan_destroy.argtypes = [vp]
        L.qk_avgdown_plan_num_items.argtypes = [vp]
        L.qk_average_down.argtypes = [vp, vp, vp, vp, ci, ci]
        L.qk_interp_plan_create.argtypes = [vp, vp, P(Geometry), ci, ci * 3, ci, ci, vp, P(vp)]
        L.qk_interp_plan_destroy.argtypes = [vp]
        L.qk_interp_plan_num_items.argtypes = [vp]
        L.qk_interp_plan_item.argtypes = [vp, ci, P(ci), P(ci), ci * 3, ci * 3]
        L.qk_InterpFromCoarse.argtypes = [vp, vp, vp, vp, vp, C.c_double, C.c_double, ci, ci, ci]
        L.qk_fluxreg_create.argtypes = [vp, vp, P(Geometry), ci * 3, ci, ci, vp, ci, P(vp)]
        L.qk_fluxreg_destroy.argtypes = [vp]
        L.qk_fluxreg_num_items.argtypes = [vp]
        L.qk_fluxreg_item.argtypes = [vp, ci, P(ci), P(ci), P(ci), P(ci), ci * 3, ci * 3, ci * 3]
        L.qk_fluxreg_reset.argtypes = [vp, vp]
        L.qk_fluxreg_save.argtypes = [vp, vp]
        L.qk_fluxreg_restore.argtypes = [vp, vp]
        L.qk_fluxreg_CrseAdd.argtypes = [vp, vp, vp * 3, C.c_double * 3, C.c_double]
        L.qk_fluxreg_FineAdd.argtypes = [vp, vp, vp * 3, C.c_double * 3, C.c_double]
        L.qk_fluxreg_Reflux.argtypes = [vp, vp, vp]
        L.qk_fluxreg_set_state_component.argtypes = [vp, C.c_int]
        L.qk_copy_box.argtypes = [vp, vp, vp, vp, ci * 3, ci * 3, ci, ci, ci]
        L.qk_amr_tile_flags.argtypes = [vp, vp, vp, P(Box), ci, ci, vp]
        L.qk_amr_tile_flags_periodic.argtypes = [vp, vp, vp, P(Box), P(ci), ci, ci, vp]
        L.qk_amr_cluster_tiles.argtypes = [vp, ci * 3, ci, ci, ci, ci, P(Box), ci]
        L.qk_amr_cluster_berger_rigoutsos.argtypes = [vp, vp, ci * 3, ci, ci, ci, C.c_double, P(Box), ci]
        L.qk_PreInterpState.argtypes = [vp, vp, vp]
        L.qk_PostInterpState.argtypes = [vp, vp, vp]
        L.qk_fluxreg_create_crse_part.argtypes = [vp, vp, P(Geometry), ci * 3, ci, P(vp)]
        L.qk_pcopy_plan_create.argtypes = [vp, P(Geometry), ci, P(Box), P(ci), ci, ci, ci, P(Box), P(ci), ci, vp, ci, ci, P(vp)]
        L.qk_pcopy_plan_destroy.argtypes = [vp]
        L.qk_pcopy_plan_num_peers.argtypes = [vp]
        L.qk_pcopy_plan_peer.argtypes = [vp, ci, P(ci), P(C.c_int64), P(C.c_int64)]
        L.qk_pcopy_plan_num_items.argtypes = [vp, ci, ci]
        L.qk_pcopy_plan_item.argtypes = [vp, ci, ci, ci, P(ci), P(ci), ci * 3, ci * 3, ci * 3, P(C.c_int64)]
        L.qk_ParallelCopy_local.argtypes = [vp, vp, vp, vp, ci, ci, ci]
        L.qk_ParallelCopy_pack.argtypes = [vp, vp, ci, vp, ci, vp]
        L.qk_ParallelCopy_unpack.argtypes = [vp, vp, ci, vp, ci, vp, ci]
    _lib = L
    return L


# every symbol include/quokka_amd.h declares (checked by the CPU test-suite without a GPU)
BOXES_ALL, BOXES_LOCAL_ONLY, BOXES_REMOTE_DEPENDENT = 0, 1, 2
TAG_CLEAR, TAG_BUF, TAG_SET = 0, 1, 2
TAGFIELD_PRESSURE = -1

DECLARED_SYMBOLS = [
    "qk_ctx_create", "qk_ctx_destroy", "qk_last_error", "qk_version", "qk_level_create", "qk_level_destroy",
    "qk_upload_array4_table", "qk_upload_iarray4_table", "qk_clear_bytes", "qk_profile_enable", "qk_profile_only", "qk_profile_reset", "qk_profile_num_kernels", "qk_profile_get",
    "qk_ReconstructStatesConstant", "qk_ReconstructStatesPLM", "qk_ReconstructStatesPPM",
    "qk_hydro_ConservedToPrimitive", "qk_hydro_ComputeFlatteningCoefficients", "qk_hydro_FlattenShocks",
    "qk_hydro_ComputeFluxes", "qk_hydro_ComputeRhsFromFluxes", "qk_hydro_AddInternalEnergyPdV", "qk_hydro_PredictStep",
    "qk_hydro_EnforceLimits", "qk_hydro_SyncDualEnergy", "qk_hydro_ComputeMaxSignalSpeed", "qk_hydro_maxSignalSpeedLocal",
    "qk_replaceFluxes", "qk_Saxpy", "qk_hydro_FixupState", "qk_hydro_stage_scratch_bytes", "qk_hydro_stage_fused",
    "qk_rad_ConservedToPrimitive", "qk_rad_ComputeWavespeedCorrection", "qk_rad_ComputeFluxes", "qk_rad_computeRadiationFluxes", "qk_rad_PredictStep", "qk_rad_AddFluxesRK2", "qk_rad_stage_fused",
    "qk_rad_AddSourceTermsSingleGroup", "qk_rad_AddSourceTermsSingleGroupMirror", "qk_rad_AddSourceTermsMultiGroup", "qk_rad_mg_planck_fractions",
    "qk_ghost_plan_create", "qk_ghost_plan_destroy", "qk_ghost_plan_num_peers", "qk_ghost_plan_peer", "qk_ghost_plan_num_items", "qk_ghost_plan_item",
    "qk_FillBoundary_local", "qk_FillBoundary_local_int", "qk_FillBoundary_pack", "qk_FillBoundary_unpack", "qk_FillBoundary_pack_int", "qk_FillBoundary_unpack_int", "qk_SumBoundary_local", "qk_SumBoundary_pack", "qk_SumBoundary_unpack", "qk_FillPhysicalBoundary",
    "qk_FillPhysicalBoundary_subset", "qk_ghost_plan_set_components", "qk_ghost_plan_box_is_remote", "qk_ghost_plan_set_box_remote",
    "qk_tag_relative_gradient", "qk_tag_centered_gradient", "qk_avgdown_plan_create", "qk_avgdown_plan_destroy", "qk_avgdown_plan_num_items", "qk_average_down", "qk_PreInterpState", "qk_PostInterpState",
    "qk_interp_plan_create", "qk_interp_plan_destroy", "qk_interp_plan_num_items", "qk_interp_plan_item", "qk_InterpFromCoarse",
    "qk_fluxreg_create", "qk_fluxreg_destroy", "qk_fluxreg_num_items", "qk_fluxreg_item", "qk_fluxreg_reset", "qk_fluxreg_save", "qk_fluxreg_restore", "qk_fluxreg_CrseAdd", "qk_fluxreg_FineAdd",
    "qk_fluxreg_Reflux", "qk_fluxreg_set_state_component", "qk_amr_tile_flags", "qk_amr_tile_flags_periodic", "qk_amr_cluster_tiles", "qk_amr_cluster_berger_rigoutsos", "qk_copy_box",
    "qk_cloudy_tables_read", "qk_cloudy_tables_free", "qk_cooling_tabulated", "qk_cooling_evaluate",
    "qk_fluxreg_create_crse_part", "qk_pcopy_plan_create", "qk_pcopy_plan_destroy", "qk_pcopy_plan_num_peers", "qk_pcopy_plan_peer", "qk_pcopy_plan_num_items",
    "qk_pcopy_plan_item", "qk_ParallelCopy_local", "qk_ParallelCopy_pack", "qk_ParallelCopy_unpack",
]


def check(ctx, rc: int, what: str = "") -> None:
    if rc != QK_OK:
        msg = lib().qk_last_error(ctx).decode() if ctx else ""
        raise QkError(f"{what} failed with status {rc}: {msg}")
