# reference_cpu_baseline.jl — the reference's own CPU() path timed beside the GPU number (BASELINE.md §3 row 1; SURVEY.md §8d
# "CPU baseline beside it"; VERDICT r5 item 4).  NEVER EXECUTED in the build image (no Julia there): oracle/upstream_probe.py
# runs it from bench.py on the first box where `julia -e 'using ClimaOcean'` works:
#
#     JULIA_NUM_THREADS=$(nproc) julia --threads $(nproc) reference_cpu_baseline.jl <dir>
#
# <dir> holds bench.py's inputs as C-order Float64 .npy files: ocean_{T,S,u,v,mask}.npy (halo-inclusive, (ny+2h, nx+2h)),
# jra_<variable>.npy (2, 320, 640) for the nine JRA55 variables (jra55_data_staging.jl:8), shape.npy = [nx, ny, h, ñ, seconds].
# It builds the README's model (README.md:56-77: 1/4° LatitudeLongitudeGrid, ocean_simulation, a two-snapshot
# PrescribedAtmosphere on the 640×320 JRA55 grid, Radiation(0.06, 1.0), OceanSeaIceModel) on CPU(), sets the clock to ñ of the
# snapshot interval, and times update_state! — interpolate_atmosphere_state!, compute_atmosphere_ocean_fluxes!,
# compute_net_ocean_fluxes! — for about `seconds`.  Prints ONE JSON line: {"seconds_per_pass": best, "passes": n, "threads": t}.
# Names follow ClimaOcean v0.8–0.10 / NumericalEarth 0.4–0.8, as in oracle_dump.jl; adjust both together.
using ClimaOcean, Oceananigans
using Oceananigans.Units

include(joinpath(@__DIR__, "npy_io.jl"))     # read_npy

dir = ARGS[1]
nx, ny, h, tf, seconds = read_npy(joinpath(dir, "shape.npy"))
Nx, Ny, H = Int(nx), Int(ny), Int(h)
grid = LatitudeLongitudeGrid(CPU(); size = (Nx, Ny, 10), halo = (H, H, H), longitude = (0, 360), latitude = (-70, 70), z = (-1000, 0))   # README.md:56-61
mask = read_npy(joinpath(dir, "ocean_mask.npy"))
inner(A) = permutedims(A[H+1:H+Ny, H+1:H+Nx])
grid = ImmersedBoundaryGrid(grid, GridFittedBottom(ifelse.(inner(mask) .> 0, -1000.0, 10.0)))
ocean = ocean_simulation(grid)                                                                                                          # README.md:67
for (f, k) in ((ocean.model.tracers.T, "T"), (ocean.model.tracers.S, "S"), (ocean.model.velocities.u, "u"), (ocean.model.velocities.v, "v"))
    src = permutedims(read_npy(joinpath(dir, "ocean_$k.npy")))
    P = parent(f)
    P[1:size(src, 1), 1:size(src, 2), size(P, 3) - H] .= src          # the surface level k = Nz, halos included
end

sgrid = LatitudeLongitudeGrid(CPU(), Float32; size = (640, 320, 1), halo = (3, 3, 1), z = (0, 1), longitude = (-0.28125, 359.71875),
                              latitude = (-89.57 - 0.2808, 89.57 + 0.2808), topology = (Periodic, Bounded, Bounded))
atmosphere = PrescribedAtmosphere(sgrid, [0.0, 3hours])
jra(v, n) = permutedims(read_npy(joinpath(dir, "jra_$(v).npy"))[n, :, :])
for n in 1:2
    Oceananigans.interior(atmosphere.tracers.T[n], :, :, 1) .= jra("tas", n);    Oceananigans.interior(atmosphere.tracers.q[n], :, :, 1) .= jra("huss", n)
    Oceananigans.interior(atmosphere.pressure[n], :, :, 1) .= jra("psl", n)
    Oceananigans.interior(atmosphere.velocities.u[n], :, :, 1) .= jra("uas", n); Oceananigans.interior(atmosphere.velocities.v[n], :, :, 1) .= jra("vas", n)
    Oceananigans.interior(atmosphere.downwelling_radiation.longwave[n], :, :, 1) .= jra("rlds", n)
    Oceananigans.interior(atmosphere.downwelling_radiation.shortwave[n], :, :, 1) .= jra("rsds", n)
    Oceananigans.interior(atmosphere.freshwater_flux.rain[n], :, :, 1) .= jra("prra", n)
    Oceananigans.interior(atmosphere.freshwater_flux.snow[n], :, :, 1) .= jra("prsn", n)
end
radiation = Radiation(ocean_albedo = 0.06, ocean_emissivity = 1.0)                                                                      # atmosphere.jl:41-44
model = OceanSeaIceModel(ocean; atmosphere, radiation)                                                                                  # README.md:75
model.clock.time = tf * 3hours

ClimaOcean.update_state!(model)          # compile + first touch
best, passes, t_start = Inf, 0, time()
while passes < 3 || (time() - t_start < seconds && passes < 100)
    t = @elapsed ClimaOcean.update_state!(model)
    global best = min(best, t); global passes += 1
end
println("{\"seconds_per_pass\": $(best), \"passes\": $(passes), \"threads\": $(Threads.nthreads())}")
