"""The Julia binding (climaocean.jl_amd/julia/CoFluxMI355X.jl) cannot be executed here (no Julia in the image), so it is
held to the C ABI textually: every `ccall` names a symbol the library exports, every function the header declares has a
binding (or is listed as deliberately host-only), and every struct mirror has the size of its ctypes twin — the header,
abi.py and the stub are three hand-kept copies of one layout."""
import ctypes as C
import os
import re

from coflux import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = open(os.path.join(ROOT, "climaocean.jl_amd", "julia", "CoFluxMI355X.jl")).read()
HEADER = open(os.path.join(ROOT, "include", "coflux.h")).read()

# measurement / self-test hooks a Julia host has no use for
NOT_BOUND = {"cf_version", "cf_set_flux_params", "cf_set_stream", "cf_debug_eval", "cf_debug_chunk_plan", "cf_device_free",
             "cf_time_stage", "cf_time_copy", "cf_profile_enable", "cf_profile_read", "cf_comm_destroy", "cf_window_upload", "cf_solver_path"}

JL_SIZE = {"Int32": 4, "Cint": 4, "Int64": 8, "Float64": 8, "Float32": 4}
TWINS = {"CfGrid": abi.Grid, "CfRoughness": abi.Roughness, "CfThermodynamics": abi.Thermodynamics, "CfSeawater": abi.Seawater,
         "CfFluxParams": abi.FluxParams, "CfOceanSurface": abi.OceanSurface, "CfExchangeFields": abi.ExchangeFields,
         "CfInterfaceFluxes": abi.InterfaceFluxes, "CfSeaIceFields": abi.SeaIceFields, "CfNetOceanFluxes": abi.NetOceanFluxes,
         "CfAtmosSource": abi.AtmosSource, "CfInterpWeights": abi.InterpWeights, "CfSeaIceParams": abi.SeaIceParams,
         "CfSeaIceState": abi.SeaIceState, "CfNetSeaIceFluxes": abi.NetSeaIceFluxes, "CfRunSchedule": abi.RunSchedule,
         "CfSeaIceAlbedoParams": abi.SeaIceAlbedoParams, "CfIceOceanParams": abi.IceOceanParams, "CfIceOceanFluxes": abi.IceOceanFluxes,
         "CfLandSource": abi.LandSource}


def julia_structs():
    out = {}
    field = r"(\w+)::((?:NTuple\{\d+,\s*(?:Ptr\{\w+\}|\w+)\})|(?:Ptr\{[^}]*\})|\w+)"
    # a struct ends at a line `end` or at `; end` closing its last line
    for m in re.finditer(r"^(?:mutable )?struct (Cf\w+)\b(.*?)(?:^end\b|; end[ \t]*(?:#.*)?$)", STUB, re.S | re.M):
        out[m.group(1)] = re.findall(field, re.sub(r"#.*", "", m.group(2)))
    return out


def jl_sizeof(t, structs):
    if t.startswith("Ptr{"):
        return 8, 8
    m = re.match(r"NTuple\{(\d+),\s*(Ptr\{\w+\}|\w+)\}", t)
    if m:
        sz, al = jl_sizeof(m.group(2), structs)
        return int(m.group(1)) * sz, al
    if t in JL_SIZE:
        return JL_SIZE[t], JL_SIZE[t]
    return struct_size(t, structs)


def struct_size(name, structs):
    off, align = 0, 1
    for _f, t in structs[name]:
        sz, al = jl_sizeof(t, structs)
        off = (off + al - 1) // al * al + sz
        align = max(align, al)
    return (off + align - 1) // align * align, align


def test_every_ccall_names_an_exported_symbol_and_every_entry_point_is_bound():
    lib = abi.load_library()
    called = set(re.findall(r"\(:(cf_\w+), libcoflux\)", STUB))
    for name in called:
        assert hasattr(lib, name), f"the stub calls {name}, which libcoflux does not export"
    declared = set(re.findall(r"\b(cf_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)))
    declared = {d for d in declared if d in abi.EXPORTED_SYMBOLS}
    assert declared == set(abi.EXPORTED_SYMBOLS), sorted(set(abi.EXPORTED_SYMBOLS) ^ declared)
    missing = declared - called - NOT_BOUND
    assert not missing, f"declared in include/coflux.h but not bound in CoFluxMI355X.jl: {sorted(missing)}"


def test_struct_mirrors_have_the_abi_sizes():
    structs = julia_structs()
    for name, twin in TWINS.items():
        assert name in structs, f"{name} missing from the stub"
        assert struct_size(name, structs)[0] == C.sizeof(twin), (name, struct_size(name, structs)[0], C.sizeof(twin))
        assert len(structs[name]) >= 1
    # field NAMES of the flat pointer bundles line up with the ctypes mirrors (order is the ABI)
    for name in ("CfOceanSurface", "CfExchangeFields", "CfInterfaceFluxes", "CfSeaIceFields", "CfNetOceanFluxes", "CfSeaIceState",
                 "CfIceOceanFluxes", "CfRunSchedule", "CfLandSource", "CfSeaIceAlbedoParams", "CfIceOceanParams", "CfSeaIceParams"):
        assert [f for f, _t in structs[name]] == [f for f, *_ in TWINS[name]._fields_], name
