// Host-visible launchers of the gfx950 kernels (defined in coflux_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "coflux_device.hpp"

namespace coflux {

hipError_t launch_interpolate(hipStream_t st, const GridDesc& G, const cf_atmos_source* s, const cf_interp_weights* w,
                              const cf_exchange_fields* e, int cap);
hipError_t launch_ao_fluxes(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                            const cf_exchange_fields* e, const cf_interface_fluxes* f);
hipError_t launch_fused(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_atmos_source* s,
                        const cf_interp_weights* w, const cf_ocean_surface* o, const cf_exchange_fields* e,
                        const cf_interface_fluxes* f, int cap);
hipError_t launch_net_fluxes(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                             const cf_exchange_fields* e, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                             const cf_interp_weights* w, const cf_net_ocean_fluxes* n);
hipError_t launch_copy(hipStream_t st, void* dst, const void* src, size_t bytes);

}  // namespace coflux
