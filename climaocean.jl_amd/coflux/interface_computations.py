"""Host-side mirror of NumericalEarth.EarthSystemModels.InterfaceComputations — the types the
reference tree constructs to configure the flux path (src/OMIPConfigurations/omip_simulation.jl:14-25,
:40-113).  Same names, same keyword meaning; `flux_params(...)` lowers a configuration to the
POD block the C ABI takes (include/coflux.h: cf_flux_params).

Defaults marked UNVERIFIED are recollections of the un-vendored package;
everything the reference tree itself states is cited.
"""
from dataclasses import dataclass, field
from typing import Optional, Union

from . import abi

default_gravitational_acceleration = 9.81  # UNVERIFIED upstream default


# ---------------------------------------------------------------------------------------------
# viscosity, roughness lengths (omip_simulation.jl:41-49)
# ---------------------------------------------------------------------------------------------
@dataclass
class TemperatureDependentAirViscosity:
    """ν(T) cubic in °C (COARE 3.6); omip_simulation.jl:41."""
    C0: float = 1.326e-5
    C1: float = 1.326e-5 * 6.542e-3
    C2: float = 1.326e-5 * 8.301e-6
    C3: float = -1.326e-5 * 4.84e-9

    def coefficients(self):
        return abi.VISCOSITY_TEMPERATURE_DEPENDENT, (self.C0, self.C1, self.C2, self.C3)


@dataclass
class ConstantAirViscosity:
    nu: float = 1.5e-5

    def coefficients(self):
        return abi.VISCOSITY_CONSTANT, (self.nu, 0.0, 0.0, 0.0)


@dataclass
class WindDependentWaveFormulation:
    """Edson et al. (2013) eq. 13 wind-dependent Charnock parameter, omip_simulation.jl:35,46:
    α = max(minimum, a1·min(U, umax) + a2) (COARE 3.5/3.6 coefficients; the floor keeps the
    roughness length positive at U < 2.9 m/s where the linear fit goes negative)."""
    a1: float = 0.0017
    a2: float = -0.005
    umax: float = 19.0
    minimum: float = 0.0


@dataclass
class MomentumRoughnessLength:
    """MomentumRoughnessLength(FT; wave_formulation, air_kinematic_viscosity), omip_simulation.jl:45-47.
    A float `wave_formulation` is a constant Charnock parameter (0.02: omip_simulation.jl:263)."""
    wave_formulation: Union[float, WindDependentWaveFormulation] = 0.02
    air_kinematic_viscosity: object = field(default_factory=TemperatureDependentAirViscosity)
    gravitational_acceleration: float = default_gravitational_acceleration
    laminar_parameter: float = 0.11
    maximum_roughness_length: float = 1.0


@dataclass
class ReynoldsScalingFunction:
    A: float = 5.85e-5
    b: float = 0.72


@dataclass
class ScalarRoughnessLength:
    """ScalarRoughnessLength(FT; air_kinematic_viscosity), omip_simulation.jl:48-49."""
    air_kinematic_viscosity: object = field(default_factory=TemperatureDependentAirViscosity)
    reynolds_number_scaling_function: ReynoldsScalingFunction = field(default_factory=ReynoldsScalingFunction)
    maximum_roughness_length: float = 1.6e-4


# ---------------------------------------------------------------------------------------------
# similarity forms, stability functions, stop criteria, velocity difference
# ---------------------------------------------------------------------------------------------
class LogarithmicSimilarityProfile:
    code = abi.SIMILARITY_LOGARITHMIC


class COARELogarithmicSimilarityProfile:
    """No ψ(ℓ/L) term; omip_simulation.jl:36,43."""
    code = abi.SIMILARITY_COARE_LOGARITHMIC


@dataclass(frozen=True)
class StabilityFunctions:
    code: int
    name: str


def atmosphere_ocean_stability_functions(FT=float):
    """Edson et al. 2013 (docs/climaocean.bib:1-10)."""
    return StabilityFunctions(abi.STABILITY_EDSON2013, "edson2013")


def atmosphere_sea_ice_stability_functions(FT=float):
    """SHEBA: Paulson unstable + Grachev et al. 2007 stable; omip_simulation.jl:56,64."""
    return StabilityFunctions(abi.STABILITY_SHEBA, "sheba")


def large_yeager_stability_functions(FT=float):
    """Paulson (1970) + linear stable (−5ζ); omip_simulation.jl:96,107."""
    return StabilityFunctions(abi.STABILITY_LARGE_YEAGER, "large_yeager")


@dataclass
class ConvergenceStopCriteria:
    tolerance: float = 1e-8
    maxiter: int = 100


@dataclass
class FixedIterations:
    """FixedIterations(5), omip_simulation.jl:22,89."""
    iterations: int = 5


class RelativeVelocity:
    """Δu = u_atm − u_ocean (OMIP-2 α=1); omip_simulation.jl:135."""
    code = abi.VELOCITY_RELATIVE


class WindVelocity:
    """Δu = u_atm; omip_simulation.jl:136."""
    code = abi.VELOCITY_WIND


# ---------------------------------------------------------------------------------------------
# SimilarityTheoryFluxes (omip_simulation.jl:42-49, 63-69, 106-113)
# ---------------------------------------------------------------------------------------------
@dataclass
class SimilarityTheoryFluxes:
    von_karman_constant: float = 0.4
    gustiness_parameter: float = 1.0
    minimum_gustiness: float = 0.2          # UNVERIFIED default (COARE); 0.5 in ":corrected", :40,44
    shear_gustiness_coefficient: float = 0.0   # c of the shear-aware form (launch.sh:67-72): 0 = off, see include/coflux.h
    stability_functions: StabilityFunctions = field(default_factory=atmosphere_ocean_stability_functions)
    momentum_roughness_length: Union[float, MomentumRoughnessLength] = field(default_factory=MomentumRoughnessLength)
    temperature_roughness_length: Union[float, ScalarRoughnessLength] = field(default_factory=ScalarRoughnessLength)
    water_vapor_roughness_length: Union[float, ScalarRoughnessLength] = field(default_factory=ScalarRoughnessLength)
    similarity_form: object = field(default_factory=LogarithmicSimilarityProfile)
    solver_stop_criteria: object = field(default_factory=ConvergenceStopCriteria)
    similarity_profile_floor: float = 1.0   # restatement guard, see include/coflux.h


@dataclass
class LargeYeagerTransferCoefficients:
    """LargeYeagerTransferCoefficients(FT), omip_simulation.jl:88: neutral 10-m coefficients of Large &
    Yeager (2004, 2009): 10³·Cd_N10 = 2.70/U + 0.142 + 0.0764·U − 3.14807e-10·U⁶ (= 2.34 for U ≥ 33 m/s),
    10³·Ce_N10 = 34.6·√Cd_N10, 10³·Ch_N10 = 18.0·√Cd_N10 (stable) or 32.7·√Cd_N10 (unstable)."""
    cd: tuple = (2.70, 0.142, 0.0764, -3.14807e-10)
    high_wind: float = 33.0
    cd_high: float = 2.34
    ce: float = 34.6
    ch_stable: float = 18.0
    ch_unstable: float = 32.7
    minimum_wind: float = 0.5
    zeta_bound: float = 10.0


@dataclass
class CoefficientBasedFluxes:
    """CoefficientBasedFluxes(FT; transfer_coefficients, solver_stop_criteria), omip_simulation.jl:86-89."""
    transfer_coefficients: LargeYeagerTransferCoefficients = field(default_factory=LargeYeagerTransferCoefficients)
    solver_stop_criteria: object = field(default_factory=lambda: FixedIterations(5))
    von_karman_constant: float = 0.4


def ncar_atmosphere_ocean_fluxes(FT=float):
    """omip_simulation.jl:79-89: OMIP-2 standard Large & Yeager bulk algorithm, 5 fixed iterations."""
    return CoefficientBasedFluxes(transfer_coefficients=LargeYeagerTransferCoefficients(),
                                  solver_stop_criteria=FixedIterations(5))


def corrected_atmosphere_ocean_fluxes(FT=float, minimum_gustiness=0.5):
    """omip_simulation.jl:40-50."""
    nu = TemperatureDependentAirViscosity()
    return SimilarityTheoryFluxes(
        similarity_form=COARELogarithmicSimilarityProfile(),
        minimum_gustiness=minimum_gustiness,
        momentum_roughness_length=MomentumRoughnessLength(
            wave_formulation=WindDependentWaveFormulation(), air_kinematic_viscosity=nu),
        temperature_roughness_length=ScalarRoughnessLength(air_kinematic_viscosity=nu),
        water_vapor_roughness_length=ScalarRoughnessLength(air_kinematic_viscosity=nu))


def shear_aware_atmosphere_ocean_fluxes(FT=float, shear_gustiness_coefficient=0.04, minimum_gustiness=0.5):
    """The `:shear_aware` flux configuration the reference's launcher describes (experiments/OMIPSimulations/scripts/
    launch.sh:67-72, emitted at :350): the `:corrected` fluxes with the Mahrt–Sun (1995) / Edson (2013) gustiness
    U_G² = (β w★)² + (c |Δu|)² + U_G,0², c = 0.04.  The reference's build_coupled_model does not accept the symbol
    (omip_simulation.jl:160); models.build_coupled_model takes it only with allow_shear_aware=True."""
    f = corrected_atmosphere_ocean_fluxes(FT, minimum_gustiness=minimum_gustiness)
    f.shear_gustiness_coefficient = float(shear_gustiness_coefficient)
    return f


def corrected_atmosphere_sea_ice_fluxes(FT=float):
    """omip_simulation.jl:62-69."""
    return SimilarityTheoryFluxes(
        stability_functions=atmosphere_sea_ice_stability_functions(),
        similarity_form=COARELogarithmicSimilarityProfile(),
        minimum_gustiness=0.2,
        momentum_roughness_length=5e-4,
        temperature_roughness_length=5e-5,
        water_vapor_roughness_length=5e-5)


def ncar_atmosphere_sea_ice_fluxes(FT=float):
    """omip_simulation.jl:105-113."""
    return SimilarityTheoryFluxes(
        stability_functions=large_yeager_stability_functions(),
        similarity_form=COARELogarithmicSimilarityProfile(),
        gustiness_parameter=0.0,
        minimum_gustiness=0.5,
        momentum_roughness_length=5e-4,
        temperature_roughness_length=5e-4,
        water_vapor_roughness_length=5e-4)


# ---------------------------------------------------------------------------------------------
# thermodynamics / ocean / radiation property blocks
# ---------------------------------------------------------------------------------------------
@dataclass
class AtmosphereThermodynamicsParameters:
    gas_constant: float = 8.3144598
    dry_air_molar_mass: float = 0.02897
    water_molar_mass: float = 0.018015
    dry_air_adiabatic_exponent: float = 2.0 / 7.0
    water_vapor_heat_capacity: float = 1859.0
    liquid_water_heat_capacity: float = 4181.0
    water_ice_heat_capacity: float = 2100.0
    reference_vaporization_enthalpy: float = 2500800.0
    reference_sublimation_enthalpy: float = 2834400.0
    reference_temperature: float = 273.16
    triple_point_temperature: float = 273.16
    triple_point_pressure: float = 611.657
    water_freezing_temperature: float = 273.15
    total_ice_nucleation_temperature: float = 233.0
    ice_nucleation_power: float = 1.0


@dataclass
class SeawaterComposition:
    water_molar_mass: float = 18.02
    constituent_molar_mass: tuple = (35.45, 22.99, 96.06, 24.31)       # Cl, Na, SO4, Mg
    constituent_mass_fraction: tuple = (0.56, 0.31, 0.08, 0.05)


@dataclass
class OceanProperties:
    reference_density: float = 1026.0             # visualize/common.jl:17
    heat_capacity: float = 3991.86795711963       # visualize/common.jl:18
    freshwater_density: float = 1000.0
    temperature_offset: float = 273.15            # ocean T in °C
    surface_z: float = -150.0                     # z of the top cell centre (README: 3000 m / 10 levels)


@dataclass
class SurfaceRadiationProperties:
    """SurfaceRadiationProperties(albedo, emissivity); atmosphere.jl:43-44."""
    albedo: object = 0.06
    emissivity: float = 1.0


@dataclass
class LatitudeDependentAlbedo:
    """α = diffuse − direct·cos(2φ) (Large & Yeager 2009)."""
    diffuse: float = 0.069
    direct: float = 0.011


@dataclass
class SeaIceInterfaceProperties:
    """SkinTemperature(ConductiveFlux) + SurfaceRadiationProperties(sea_ice_albedo, 1.0) of the
    atmosphere–sea-ice interface (atmosphere.jl:34-44).  Values marked UNVERIFIED are ClimaSeaIce /
    NumericalEarth defaults as recalled."""
    conductivity: float = 2.0                    # UNVERIFIED
    consolidation_thickness: float = 0.05        # UNVERIFIED
    maximum_temperature_change: float = 5.0      # UNVERIFIED
    ice_salinity: float = 4.0                    # UNVERIFIED
    liquidus_slope: float = 0.054
    freshwater_melting_temperature: float = 273.15
    albedo: float = 0.7
    emissivity: float = 1.0                      # atmosphere.jl:44
    temperature_offset: float = 273.15
    skin_temperature_scheme: int = 0             # abi.SKIN_EXPLICIT (as recalled) / abi.SKIN_SEMI_IMPLICIT (damped form)

    def to_params(self):
        import ctypes
        p = abi.SeaIceParams()
        p.struct_size = ctypes.sizeof(abi.SeaIceParams)
        for name in ("conductivity", "consolidation_thickness", "maximum_temperature_change", "ice_salinity",
                     "liquidus_slope", "freshwater_melting_temperature", "albedo", "emissivity", "temperature_offset",
                     "skin_temperature_scheme"):
            setattr(p, name, getattr(self, name))
        return p


@dataclass
class MomentumBasedFrictionVelocity:
    """u★ from the actual ice–ocean stress (omip_simulation.jl:74-77)."""
    minimum: float = 0.0


@dataclass
class ThreeEquationHeatFlux:
    """ThreeEquationHeatFlux(; friction_velocity = MomentumBasedFrictionVelocity()) — corrected_ice_ocean_heat_flux(),
    omip_simulation.jl:71-77.  Coefficients are the recalled ClimaSeaIce defaults (UNVERIFIED)."""
    friction_velocity: MomentumBasedFrictionVelocity = field(default_factory=MomentumBasedFrictionVelocity)
    heat_transfer_coefficient: float = 0.0095
    salt_transfer_coefficient: float = 0.0095 / 35.0
    ice_density: float = 917.0
    latent_heat_of_fusion: float = 334000.0
    ice_salinity: float = 4.0
    liquidus_slope: float = 0.054

    def to_params(self, top_cell_thickness, time_step):
        import ctypes
        p = abi.IceOceanParams()
        p.struct_size = ctypes.sizeof(abi.IceOceanParams)
        for name in ("heat_transfer_coefficient", "salt_transfer_coefficient", "ice_density", "latent_heat_of_fusion",
                     "ice_salinity", "liquidus_slope"):
            setattr(p, name, getattr(self, name))
        p.minimum_friction_velocity = self.friction_velocity.minimum
        p.top_cell_thickness, p.time_step = float(top_cell_thickness), float(time_step)
        return p


def corrected_ice_ocean_heat_flux():
    """omip_simulation.jl:77."""
    return ThreeEquationHeatFlux()


@dataclass
class SeaIceAlbedo:
    """SeaIceAlbedo(hi, hs, Ts) — the CCSM3 albedo of atmosphere.jl:30-44; it reads the sea-ice model's live thickness,
    snow thickness and top temperature, which here are the PrescribedSeaIce fields."""
    ice_visible: float = 0.78
    ice_near_infrared: float = 0.36
    snow_visible: float = 0.98
    snow_near_infrared: float = 0.70
    ocean_albedo: float = 0.06
    reference_thickness: float = 0.3
    melt_temperature_range: float = 1.0
    ice_melt_change: float = 0.075
    snow_melt_change_visible: float = 0.10
    snow_melt_change_near_infrared: float = 0.15
    snow_patch_thickness: float = 0.02
    visible_fraction: float = 0.5
    melting_temperature: float = 0.0

    def to_params(self):
        import ctypes
        p = abi.SeaIceAlbedoParams()
        p.struct_size = ctypes.sizeof(abi.SeaIceAlbedoParams)
        for f in self.__dataclass_fields__:
            setattr(p, f, getattr(self, f))
        return p


def _roughness_block(r, scalar):
    b = abi.Roughness()
    if isinstance(r, (int, float)):
        b.kind = abi.SCALAR_ROUGHNESS_CONSTANT if scalar else abi.ROUGHNESS_CONSTANT
        b.constant_length = float(r)
        b.maximum_length = float(r)
        b.viscosity_kind = abi.VISCOSITY_CONSTANT
        b.viscosity[0] = 1.5e-5
        return b
    vk, coef = r.air_kinematic_viscosity.coefficients()
    b.viscosity_kind = vk
    for k in range(4):
        b.viscosity[k] = coef[k]
    b.maximum_length = r.maximum_roughness_length
    if scalar:
        b.kind = abi.SCALAR_ROUGHNESS_REYNOLDS
        b.reynolds_A = r.reynolds_number_scaling_function.A
        b.reynolds_b = r.reynolds_number_scaling_function.b
    else:
        b.laminar = r.laminar_parameter
        if isinstance(r.wave_formulation, WindDependentWaveFormulation):
            b.kind = abi.ROUGHNESS_WIND_CHARNOCK
            b.wind_a1 = r.wave_formulation.a1
            b.wind_a2 = r.wave_formulation.a2
            b.wind_umax = r.wave_formulation.umax
            b.charnock = r.wave_formulation.minimum
        else:
            b.kind = abi.ROUGHNESS_CHARNOCK
            b.charnock = float(r.wave_formulation)
    return b


def flux_params(fluxes: Optional[SimilarityTheoryFluxes] = None, *,
                velocity_difference=None,
                thermodynamics: Optional[AtmosphereThermodynamicsParameters] = None,
                seawater: Optional[SeawaterComposition] = None,
                ocean: Optional[OceanProperties] = None,
                ocean_surface: Optional[SurfaceRadiationProperties] = None,
                reference_height=10.0, boundary_layer_height=600.0,
                gravitational_acceleration=default_gravitational_acceleration,
                ocean_minimum_salinity=0.0, stefan_boltzmann_constant=5.67e-8,
                mask_kind=abi.MASK_U8, penetrating_shortwave=True) -> abi.FluxParams:
    """Lower a flux configuration to the C ABI's cf_flux_params block."""
    ly = None
    if isinstance(fluxes, CoefficientBasedFluxes):
        ly = fluxes
        # the similarity block still has to be valid; Paulson/−5ζ are the Large–Yeager stability functions
        fluxes = SimilarityTheoryFluxes(von_karman_constant=ly.von_karman_constant,
                                        stability_functions=large_yeager_stability_functions(),
                                        similarity_form=COARELogarithmicSimilarityProfile(),
                                        solver_stop_criteria=ly.solver_stop_criteria)
    f = fluxes or SimilarityTheoryFluxes()
    th = thermodynamics or AtmosphereThermodynamicsParameters()
    sw = seawater or SeawaterComposition()
    oc = ocean or OceanProperties()
    rad = ocean_surface or SurfaceRadiationProperties()
    vd = velocity_difference or RelativeVelocity()

    p = abi.FluxParams()
    import ctypes
    p.struct_size = ctypes.sizeof(abi.FluxParams)
    p.abi_version = abi.ABI_VERSION
    p.similarity_form = f.similarity_form.code
    p.stability_functions = f.stability_functions.code
    sc = f.solver_stop_criteria
    if isinstance(sc, FixedIterations):
        p.stop_kind, p.maxiter, p.tolerance = abi.STOP_FIXED, sc.iterations, 0.0
    else:
        p.stop_kind, p.maxiter, p.tolerance = abi.STOP_CONVERGENCE, sc.maxiter, sc.tolerance
    p.velocity_difference = vd.code
    p.mask_kind = mask_kind
    p.von_karman = f.von_karman_constant
    p.gustiness_parameter = f.gustiness_parameter
    p.minimum_gustiness = f.minimum_gustiness
    p.shear_gustiness_coefficient = getattr(f, "shear_gustiness_coefficient", 0.0)
    p.similarity_profile_floor = f.similarity_profile_floor
    p.momentum_roughness = _roughness_block(f.momentum_roughness_length, scalar=False)
    p.temperature_roughness = _roughness_block(f.temperature_roughness_length, scalar=True)
    p.water_vapor_roughness = _roughness_block(f.water_vapor_roughness_length, scalar=True)
    p.reference_height = reference_height
    p.boundary_layer_height = boundary_layer_height
    p.gravitational_acceleration = gravitational_acceleration
    t = p.thermo
    t.gas_constant, t.dry_air_molar_mass, t.water_molar_mass = th.gas_constant, th.dry_air_molar_mass, th.water_molar_mass
    t.kappa_d = th.dry_air_adiabatic_exponent
    t.cp_v, t.cp_l, t.cp_i = th.water_vapor_heat_capacity, th.liquid_water_heat_capacity, th.water_ice_heat_capacity
    t.LH_v0, t.LH_s0 = th.reference_vaporization_enthalpy, th.reference_sublimation_enthalpy
    t.T_0, t.T_triple, t.p_triple = th.reference_temperature, th.triple_point_temperature, th.triple_point_pressure
    t.T_freeze, t.T_icenuc, t.pow_icenuc = th.water_freezing_temperature, th.total_ice_nucleation_temperature, th.ice_nucleation_power
    p.seawater.water_molar_mass = sw.water_molar_mass
    for k in range(4):
        p.seawater.constituent_molar_mass[k] = sw.constituent_molar_mass[k]
        p.seawater.constituent_mass_fraction[k] = sw.constituent_mass_fraction[k]
    p.ocean_reference_density = oc.reference_density
    p.ocean_heat_capacity = oc.heat_capacity
    p.ocean_freshwater_density = oc.freshwater_density
    p.ocean_temperature_offset = oc.temperature_offset
    p.ocean_minimum_salinity = ocean_minimum_salinity
    p.ocean_surface_z = oc.surface_z
    if isinstance(rad.albedo, LatitudeDependentAlbedo):
        p.ocean_albedo_kind = abi.ALBEDO_LATITUDE_DEPENDENT
        p.ocean_albedo_diffuse = rad.albedo.diffuse
        p.ocean_albedo_direct = rad.albedo.direct
        p.ocean_albedo = rad.albedo.diffuse
    else:
        p.ocean_albedo_kind = abi.ALBEDO_CONSTANT
        p.ocean_albedo = float(rad.albedo)
        p.ocean_albedo_diffuse, p.ocean_albedo_direct = 0.069, 0.011
    p.penetrating_shortwave = 1 if penetrating_shortwave else 0
    p.ocean_emissivity = rad.emissivity
    p.stefan_boltzmann = stefan_boltzmann_constant
    tc = ly.transfer_coefficients if ly is not None else LargeYeagerTransferCoefficients()
    p.flux_formulation = abi.FORMULATION_LARGE_YEAGER if ly is not None else abi.FORMULATION_SIMILARITY
    p.ly_minimum_wind, p.ly_zeta_bound = tc.minimum_wind, tc.zeta_bound
    for k in range(4):
        p.ly_cd[k] = tc.cd[k]
    p.ly_high_wind, p.ly_cd_high = tc.high_wind, tc.cd_high
    p.ly_ce, p.ly_ch_stable, p.ly_ch_unstable = tc.ce, tc.ch_stable, tc.ch_unstable
    return p
