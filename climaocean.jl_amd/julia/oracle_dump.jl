# oracle_dump.jl — the upstream pin (SURVEY.md §7-H1, §8c).  NEVER EXECUTED in the build image (no Julia there);
# run it on any box where `julia -e 'using ClimaOcean, NumericalEarth'` succeeds:
#
#     julia --project=<env with ClimaOcean> climaocean.jl_amd/julia/oracle_dump.jl
#
# It feeds the committed synthetic inputs of tests/golden/upstream_inputs/*.npy (the 24×12 tile of
# tests/golden/flux_path_24x12.npz: ocean surface state + the atmosphere state already on the ocean grid) through the
# reference's PUBLIC API — ocean_simulation, PrescribedAtmosphere, Radiation, OceanSeaIceModel (README.md:67-75,
# src/ClimaOcean.jl:31-42) — for every flux formulation the tree configures (src/OMIPConfigurations/omip_simulation.jl:40-113)
# and writes tests/golden/upstream/<formulation>_<field>.npy.  tests/test_upstream_pin.py then compares the CPU oracle and
# (on a GPU box) the HIP path with those files; until they exist every report says "parity unpinned".
# Field and keyword names follow ClimaOcean v0.8–0.10 / NumericalEarth 0.4–0.8; adjust here if the installed version
# renamed them — the INPUTS and the OUTPUT file names are the contract, not this glue.
using ClimaOcean, Oceananigans
using Oceananigans.Units

const ROOT = normpath(joinpath(@__DIR__, "..", ".."))
const INP = joinpath(ROOT, "tests", "golden", "upstream_inputs")
const OUT = joinpath(ROOT, "tests", "golden", "upstream")

# ---- minimal .npy (v1.0, little-endian Float64, C order) reader / writer -------------------------------------------------
function read_npy(path)
    open(path) do io
        read(io, 6) == UInt8[0x93, 'N', 'U', 'M', 'P', 'Y'] || error("not an .npy file: $path")
        read(io, 2); hlen = Int(read(io, UInt16))
        header = String(read(io, hlen))
        occursin("'<f8'", header) && occursin("'fortran_order': False", header) || error("expected C-order <f8: $header")
        dims = parse.(Int, split(strip(match(r"\(([^)]*)\)", header).captures[1], [' ', ',']), r"\s*,\s*"; keepempty = false))
        data = Vector{Float64}(undef, prod(dims)); read!(io, data)
        length(dims) == 1 ? data : permutedims(reshape(data, reverse(dims)...))   # rows = j, columns = i
    end
end

function write_npy(path, A::AbstractMatrix{Float64})   # A[j, i] → C-order (ny, nx)
    header = "{'descr': '<f8', 'fortran_order': False, 'shape': ($(size(A, 1)), $(size(A, 2))), }"
    header *= " "^(63 - (10 + length(header)) % 64) * "\n"
    open(path, "w") do io
        write(io, UInt8[0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0]); write(io, UInt16(length(header))); write(io, header)
        write(io, collect(permutedims(A)))
    end
end

# ---- inputs --------------------------------------------------------------------------------------------------------------
Nx, Ny, H, ring = Int.(read_npy(joinpath(INP, "shape.npy")))
ocean_in = Dict(k => read_npy(joinpath(INP, "ocean_$k.npy")) for k in ("T", "S", "u", "v", "mask"))
atmos_in = Dict(k => read_npy(joinpath(INP, "atmos_$k.npy")) for k in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp"))
interior(A) = permutedims(A[H+1:H+Ny, H+1:H+Nx])          # → (Nx, Ny), Oceananigans' i-fastest order
withhalo(A) = permutedims(A)                               # (Nx+2H, Ny+2H)

# The tile is a 55–58°N band of the 1/4° grid; only the index space matters (the atmosphere is already interpolated).
grid = LatitudeLongitudeGrid(CPU(); size = (Nx, Ny, 1), halo = (H, H, H), longitude = (0, Nx / 4), latitude = (55, 55 + Ny / 4),
                             z = (-10, 0), topology = (Periodic, Bounded, Bounded))
bottom = ifelse.(interior(ocean_in["mask"]) .> 0, -10.0, 10.0)        # land where the mask is 0
grid = ImmersedBoundaryGrid(grid, GridFittedBottom(bottom))
ocean = ocean_simulation(grid; momentum_advection = nothing, tracer_advection = nothing, closure = nothing)
for (f, k) in ((ocean.model.tracers.T, "T"), (ocean.model.tracers.S, "S"), (ocean.model.velocities.u, "u"), (ocean.model.velocities.v, "v"))
    P = parent(f); src = withhalo(ocean_in[k])               # halos included: the flux kernels read u[i+1], v[j+1]
    P[1:size(src, 1), 1:size(src, 2), H+1] .= src
end

atmosphere = PrescribedAtmosphere(grid, [0.0, 1.0])          # two identical snapshots ⇒ no time interpolation
for n in 1:2
    interior(atmosphere.velocities.u[n]) .= interior(atmos_in["u"]);  interior(atmosphere.velocities.v[n]) .= interior(atmos_in["v"])
    interior(atmosphere.tracers.T[n]) .= interior(atmos_in["T"]);      interior(atmosphere.tracers.q[n]) .= interior(atmos_in["q"])
    interior(atmosphere.pressure[n]) .= interior(atmos_in["p"])
    interior(atmosphere.downwelling_radiation.shortwave[n]) .= interior(atmos_in["Qs"])
    interior(atmosphere.downwelling_radiation.longwave[n]) .= interior(atmos_in["Ql"])
    interior(atmosphere.freshwater_flux.rain[n]) .= interior(atmos_in["Mp"]);  interior(atmosphere.freshwater_flux.snow[n]) .= 0
end
radiation = Radiation(ocean_albedo = 0.06, ocean_emissivity = 1.0)    # atmosphere.jl:41-44

FT = Float64
formulations = Dict(
    "default"   => SimilarityTheoryFluxes(FT),                                                        # README.md:75
    "corrected" => ClimaOcean.OMIPConfigurations.corrected_atmosphere_ocean_fluxes(FT),               # omip_simulation.jl:40-49
    "ncar"      => ClimaOcean.OMIPConfigurations.ncar_atmosphere_ocean_fluxes(FT))                    # omip_simulation.jl:79-89

mkpath(OUT)
for (name, fluxes) in formulations
    interfaces = ComponentInterfaces(atmosphere, ocean; radiation, atmosphere_ocean_fluxes = fluxes)
    model = OceanSeaIceModel(ocean; atmosphere, radiation, interfaces)                                 # runs update_state!
    ao = model.interfaces.atmosphere_ocean_interface.fluxes                                            # omip_diagnostics.jl:81-82
    net = model.interfaces.net_fluxes.ocean                                                            # omip_diagnostics.jl:77-80
    dump(field, tag) = write_npy(joinpath(OUT, "$(name)_$(tag).npy"), permutedims(Array(Oceananigans.interior(field, :, :, 1))))
    dump(ao.sensible_heat, "sensible_heat"); dump(ao.latent_heat, "latent_heat"); dump(ao.water_vapor, "water_vapor")
    dump(ao.x_momentum, "x_momentum");       dump(ao.y_momentum, "y_momentum")
    dump(net.u, "net_u"); dump(net.v, "net_v"); dump(net.T, "net_T"); dump(net.S, "net_S")
end
open(joinpath(OUT, "VERSION.txt"), "w") do io
    println(io, "ClimaOcean ", pkgversion(ClimaOcean)); println(io, "Oceananigans ", pkgversion(Oceananigans))
    isdefined(Main, :NumericalEarth) && println(io, "NumericalEarth ", pkgversion(Main.NumericalEarth))
end
@info "wrote $(OUT): copy it next to tests/golden/ and run `pytest tests/test_upstream_pin.py`"
