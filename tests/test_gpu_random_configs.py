"""Randomised flux formulations against the oracle: every combination of roughness kinds, viscosities, stability
functions, similarity forms, stop criteria, gustiness, heights and velocity differences that the parameter block
can express must take the same path on the device (including the generic, non-specialised solver) as in the
CPU restatement."""
import random

import numpy as np
import pytest

import util
from coflux import interface_computations as ic
from test_gpu_parity import compare, run_gpu, run_oracle

pytestmark = pytest.mark.gpu


def random_formulation(rng):
    visc = lambda: rng.choice([ic.TemperatureDependentAirViscosity(), ic.ConstantAirViscosity(rng.uniform(1.2e-5, 1.7e-5))])
    kind = rng.randrange(3)
    if kind == 0:
        mom = rng.choice([5e-4, 1e-4, 2e-3])
    elif kind == 1:
        mom = ic.MomentumRoughnessLength(wave_formulation=rng.choice([0.011, 0.02, 0.03]), air_kinematic_viscosity=visc(),
                                         laminar_parameter=rng.choice([0.11, 0.0]),
                                         maximum_roughness_length=rng.choice([1.0, 5e-3]))
    else:
        mom = ic.MomentumRoughnessLength(wave_formulation=ic.WindDependentWaveFormulation(minimum=rng.choice([0.0, 0.005])),
                                         air_kinematic_viscosity=visc())

    def scalar():
        if rng.random() < 0.4:
            return rng.choice([5e-5, 5e-4, 1e-5])
        return ic.ScalarRoughnessLength(air_kinematic_viscosity=visc(),
                                        reynolds_number_scaling_function=ic.ReynoldsScalingFunction(
                                            A=rng.choice([5.85e-5, 5.5e-5]), b=rng.choice([0.72, 0.6])),
                                        maximum_roughness_length=rng.choice([1.6e-4, 1.1e-4]))
    t_rough = scalar()
    q_rough = t_rough if rng.random() < 0.5 else scalar()
    stop = ic.FixedIterations(rng.choice([1, 3, 8])) if rng.random() < 0.3 else \
        ic.ConvergenceStopCriteria(tolerance=rng.choice([1e-8, 1e-6]), maxiter=rng.choice([100, 30]))
    f = ic.SimilarityTheoryFluxes(
        gustiness_parameter=rng.choice([1.0, 1.2, 0.0]), minimum_gustiness=rng.choice([0.2, 0.5, 0.0]),
        stability_functions=rng.choice([ic.atmosphere_ocean_stability_functions, ic.atmosphere_sea_ice_stability_functions,
                                        ic.large_yeager_stability_functions])(),
        momentum_roughness_length=mom, temperature_roughness_length=t_rough, water_vapor_roughness_length=q_rough,
        similarity_form=rng.choice([ic.LogarithmicSimilarityProfile, ic.COARELogarithmicSimilarityProfile])(),
        solver_stop_criteria=stop)
    vd = rng.choice([None, ic.RelativeVelocity(), ic.WindVelocity()])
    extra = dict(reference_height=rng.choice([10.0, 2.0, 20.0]), boundary_layer_height=rng.choice([600.0, 1000.0]))
    return f, vd, extra


@pytest.mark.parametrize("seed", range(24))
def test_random_formulation_matches_oracle(seed):
    rng = random.Random(1000 + seed)
    f, vd, extra = random_formulation(rng)
    if f.minimum_gustiness == 0.0 and f.gustiness_parameter == 0.0:
        f.minimum_gustiness = 0.1   # U = |Δu| can be exactly 0 otherwise: a different (degenerate) regime
    params = ic.flux_params(f, velocity_difference=vd, **extra)
    nx, ny = rng.choice([(64, 33), (97, 21), (130, 16)])
    weights = rng.choice(["latlon", "tripolar"])
    fused, use_ice = rng.random() < 0.5, rng.random() < 0.5
    case = util.build_case(nx, ny, 3, 3, weights=weights)
    got = run_gpu(case, params, fused=fused, ice=use_ice)
    ref = run_oracle(case, params, ice=use_ice)
    compare(case, got, ref, 1, maxiter=max(params.maxiter, 1))
    np.testing.assert_array_equal(
        util.window(got["fluxes"]["iterations"], 3, 3, nx, ny, 1)[util.window(ref["fluxes"]["iterations"], 3, 3, nx, ny, 1) < params.maxiter],
        util.window(ref["fluxes"]["iterations"], 3, 3, nx, ny, 1)[util.window(ref["fluxes"]["iterations"], 3, 3, nx, ny, 1) < params.maxiter])


@pytest.mark.parametrize("seed", range(12))
def test_random_radiation_and_partition_parameters_match_oracle(seed):
    """Net-flux assembly under random surface properties: constant or latitude-dependent albedo, emissivity, ocean
    density / heat capacity / freshwater density, salinity floor, shortwave to the surface flux or into JT, with
    and without the sea-ice partition, random time level pair and fraction."""
    rng = random.Random(5000 + seed)
    albedo = rng.choice([0.06, 0.1, ic.LatitudeDependentAlbedo(), ic.LatitudeDependentAlbedo(diffuse=0.08, direct=0.02)])
    props = ic.OceanProperties(reference_density=rng.choice([1026.0, 1035.0]), heat_capacity=rng.choice([3991.86795711963, 3850.0]),
                               freshwater_density=rng.choice([1000.0, 999.8]))
    params = ic.flux_params(rng.choice([ic.SimilarityTheoryFluxes(), ic.corrected_atmosphere_ocean_fluxes(),
                                        ic.ncar_atmosphere_ocean_fluxes()]),
                            ocean=props, ocean_surface=ic.SurfaceRadiationProperties(albedo, rng.choice([1.0, 0.97, 0.9])),
                            ocean_minimum_salinity=rng.choice([0.0, 1.0, 34.5]), penetrating_shortwave=rng.random() < 0.5,
                            stefan_boltzmann_constant=rng.choice([5.67e-8, 5.670374419e-8]))
    nx, ny = rng.choice([(72, 40), (101, 19)])
    case = util.build_case(nx, ny, 4, 4, weights=rng.choice(["latlon", "tripolar"]), n_levels=4)
    l1, l2 = rng.sample(range(4), 2)
    tf = rng.choice([0.0, 1.0, rng.random()])
    use_ice, fused = rng.random() < 0.6, rng.random() < 0.5
    got = run_gpu(case, params, fused=fused, ice=use_ice, time_fraction=tf, level1=l1, level2=l2)
    ref = run_oracle(case, params, ice=use_ice, time_fraction=tf, level1=l1, level2=l2)
    compare(case, got, ref, 1, maxiter=max(params.maxiter, 1))
