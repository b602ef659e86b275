/* c_abi_driver.c — exercises libcoflux exactly as a non-Python host (the Julia ccall stub) would:
 * plain C, device memory through cf_device_alloc / cf_h2d / cf_d2h, no torch.  Built and run by
 * tests/test_gpu_parity.py::test_c_driver_through_the_abi.  Covers cf_update_state, the step loop cf_time_steps (a
 * cf_run_schedule built in C), the sea-ice interface (a cf_sea_ice_state built in C) and CF_OPT_LATENCY_LAYOUT with
 * cf_solver_latency_layout.  Prints "OK <checksum>". */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/coflux.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != 0) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cf_last_error(ctx)); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void) {
    cf_ctx* ctx = NULL;
    const int nx = 96, ny = 24, h = 3, sj = nx + 2 * h, rows = ny + 2 * h;
    const size_t n = (size_t)sj * rows;
    cf_grid g = {nx, ny, h, h, 1, 0};
    cf_flux_params p;
    cf_default_flux_params(&p);
    CHECK(cf_create(&ctx, 0, &g, &p));

    /* uniform ocean and atmosphere: every wet cell must give the same fluxes */
    double *T = malloc(n * 8), *S = malloc(n * 8), *u = malloc(n * 8), *v = malloc(n * 8), *f = malloc(n * 8);
    unsigned char* mask = malloc(n);
    for (size_t k = 0; k < n; ++k) {
        T[k] = 18.0;
        S[k] = 35.0;
        u[k] = 0.1;
        v[k] = -0.05;
        mask[k] = (k % 7) != 0;
    }
    const int nsx = 640, nsy = 320;
    float* plane = malloc((size_t)2 * nsx * nsy * sizeof(float));
    const float vals[CF_JRA55_NVARS] = {288.15f, 0.008f, 101325.0f, 6.0f, 2.0f, 350.0f, 200.0f, 3e-5f, 0.0f};
    cf_atmos_source src;
    memset(&src, 0, sizeof src);
    for (int var = 0; var < CF_JRA55_NVARS; ++var) {
        for (size_t k = 0; k < (size_t)2 * nsx * nsy; ++k) plane[k] = vals[var];
        void* d = cf_device_alloc(ctx, (size_t)2 * nsx * nsy * sizeof(float));
        CHECK(cf_h2d(ctx, d, plane, (size_t)2 * nsx * nsy * sizeof(float)));
        src.data[var] = (const float*)d;
    }
    src.ns_x = nsx;
    src.ns_y = nsy;
    src.n_levels = 2;
    src.level1 = 0;
    src.level2 = 1;
    src.time_fraction = 0.25;

    double *fi = malloc(sj * 8), *fj = malloc(rows * 8);
    for (int i = 0; i < sj; ++i) fi[i] = fmod((i - h + 0.5) * 360.0 / nx + 360.0, 360.0) / (360.0 / nsx);
    for (int j = 0; j < rows; ++j) fj[j] = (-30.0 + (j - h + 0.5) * 60.0 / ny + 89.57) / (2 * 89.57 / (nsy - 1));
    cf_interp_weights w;
    memset(&w, 0, sizeof w);
    w.separable = 1;
    void *dfi = cf_device_alloc(ctx, sj * 8), *dfj = cf_device_alloc(ctx, rows * 8);
    CHECK(cf_h2d(ctx, dfi, fi, sj * 8));
    CHECK(cf_h2d(ctx, dfj, fj, rows * 8));
    w.fi = dfi;
    w.fj = dfj;

#define DEV(name, host, bytes)                   \
    void* name = cf_device_alloc(ctx, bytes);    \
    if (!name) return 2;                         \
    if (host) CHECK(cf_h2d(ctx, name, host, bytes)); 
    DEV(dT, T, n * 8) DEV(dS, S, n * 8) DEV(du, u, n * 8) DEV(dv, v, n * 8) DEV(dm, mask, n)
    cf_ocean_surface oc = {dT, dS, du, dv, dm};
    double* ex[8];
    double* fl[6];
    double* nt[5];
    memset(f, 0, n * 8);
    for (int k = 0; k < 8; ++k) { ex[k] = cf_device_alloc(ctx, n * 8); CHECK(cf_h2d(ctx, ex[k], f, n * 8)); }
    for (int k = 0; k < 6; ++k) { fl[k] = cf_device_alloc(ctx, n * 8); CHECK(cf_h2d(ctx, fl[k], f, n * 8)); }
    for (int k = 0; k < 5; ++k) { nt[k] = cf_device_alloc(ctx, n * 8); CHECK(cf_h2d(ctx, nt[k], f, n * 8)); }
    cf_exchange_fields e = {ex[0], ex[1], ex[2], ex[3], ex[4], ex[5], ex[6], ex[7]};
    cf_interface_fluxes fx;
    memset(&fx, 0, sizeof fx);
    fx.sensible_heat = fl[0]; fx.latent_heat = fl[1]; fx.water_vapor = fl[2];
    fx.x_momentum = fl[3]; fx.y_momentum = fl[4]; fx.temperature = fl[5];
    cf_net_ocean_fluxes net;
    memset(&net, 0, sizeof net);
    net.u = nt[0]; net.v = nt[1]; net.T = nt[2]; net.S = nt[3]; net.shortwave_surface_flux = nt[4];

    CHECK(cf_update_state(ctx, &src, &w, &oc, &e, &fx, NULL, &net));
    CHECK(cf_sync(ctx));
    CHECK(cf_d2h(ctx, f, fl[1], n * 8)); /* latent heat */
    double first = NAN, sum = 0;
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            size_t k = (size_t)(j + h) * sj + (i + h);
            if (!mask[k]) { if (f[k] != 0.0) { fprintf(stderr, "land cell not zero\n"); return 3; } continue; }
            if (isnan(first)) first = f[k];
            if (fabs(f[k] - first) > 1e-9 * fabs(first)) { fprintf(stderr, "non-uniform result %g vs %g\n", f[k], first); return 4; }
            sum += f[k];
        }
    if (!(first > 20.0 && first < 400.0)) { fprintf(stderr, "implausible latent heat %g\n", first); return 5; }
    /* the scheduling options of round 5 from a C caller: the latency layout forced on and off must reproduce the same bits
     * (a small surface with the default flux parameters: off in the automatic mode, cf_solver_latency_layout tells) */
    {
        int layout = -1;
        CHECK(cf_solver_latency_layout(ctx, &layout));
        if (layout != 0) { fprintf(stderr, "automatic latency layout on the logarithmic profile: %d\n", layout); return 7; }
        double* again = (double*)malloc(n * 8);
        for (int mode = 2; mode >= 0; mode -= 2) {
            CHECK(cf_set_option(ctx, CF_OPT_LATENCY_LAYOUT, mode));
            CHECK(cf_update_state(ctx, &src, &w, &oc, &e, &fx, NULL, &net));
            CHECK(cf_sync(ctx));
            CHECK(cf_solver_latency_layout(ctx, &layout));
            if (layout != (mode == 2)) { fprintf(stderr, "cf_solver_latency_layout says %d in mode %d\n", layout, mode); return 7; }
            CHECK(cf_d2h(ctx, again, fl[1], n * 8));
            if (memcmp(again, f, n * 8) != 0) { fprintf(stderr, "CF_OPT_LATENCY_LAYOUT = %d changes the latent heat's bits\n", mode); return 7; }
        }
        CHECK(cf_set_option(ctx, CF_OPT_LATENCY_LAYOUT, 1));
        if (cf_set_option(ctx, CF_OPT_LATENCY_LAYOUT, 3) == 0) { fprintf(stderr, "CF_OPT_LATENCY_LAYOUT = 3 accepted\n"); return 7; }
        free(again);
    }
    /* run!(simulation) in the library: three steps of cf_time_steps over the same uniform state must reproduce the
     * single cf_update_state above (the uniform JRA55 planes make every time fraction equivalent) */
    {
        cf_run_schedule sch;
        memset(&sch, 0, sizeof sch);
        sch.struct_size = (int)sizeof sch;
        sch.n_ocean_states = 1;
        sch.ocean_states = &oc;
        sch.n_atmos_sets = 1;
        sch.atmos = &e;
        sch.halo_backend = CF_HALO_NONE;
        sch.time_fraction = 0.1;
        sch.time_fraction_increment = 1.0 / 9.0;
        double* again = malloc(n * 8);
        memcpy(again, f, n * 8);
        CHECK(cf_h2d(ctx, fl[1], T, n * 8)); /* garbage into the output the steps must overwrite */
        CHECK(cf_time_steps(ctx, 0, 3, &sch, &src, &w, &fx, NULL, &net));
        CHECK(cf_sync(ctx));
        CHECK(cf_d2h(ctx, f, fl[1], n * 8));
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t k = (size_t)(j + h) * sj + (i + h);
                if (fabs(f[k] - again[k]) > 1e-9 * fabs(first)) { fprintf(stderr, "cf_time_steps differs at %d %d: %g vs %g\n", i, j, f[k], again[k]); return 7; }
            }
        free(again);
    }
    /* compute_atmosphere_sea_ice_fluxes!: a cf_sea_ice_state passed by a C-compiled caller; uniform thin ice under the
     * uniform atmosphere: every wet cell gets the same, finite skin temperature at or below the melting point */
    {
        cf_flux_params ip;
        cf_sea_ice_params sp;
        cf_default_flux_params(&ip);
        CHECK(cf_default_sea_ice_params(&sp));
        ip.momentum_roughness.kind = CF_ROUGHNESS_CONSTANT;
        ip.momentum_roughness.constant_length = 5e-4;
        ip.temperature_roughness.kind = CF_SCALAR_ROUGHNESS_CONSTANT;
        ip.temperature_roughness.constant_length = 5e-5;
        ip.water_vapor_roughness = ip.temperature_roughness;
        CHECK(cf_set_sea_ice_formulation(ctx, &ip, &sp));
        double *hi = malloc(n * 8), *ts = malloc(n * 8), *conc = malloc(n * 8);
        for (size_t k = 0; k < n; ++k) { hi[k] = 0.3; ts[k] = -5.0; conc[k] = 0.9; }
        DEV(dhi, hi, n * 8) DEV(dts, ts, n * 8) DEV(dconc, conc, n * 8)
        cf_sea_ice_state st;
        memset(&st, 0, sizeof st);
        st.concentration = dconc;
        st.thickness = dhi;
        st.top_temperature = dts;
        double* io[6];
        for (int k = 0; k < 6; ++k) { io[k] = cf_device_alloc(ctx, n * 8); CHECK(cf_h2d(ctx, io[k], u, n * 8)); }
        cf_interface_fluxes ix;
        memset(&ix, 0, sizeof ix);
        ix.sensible_heat = io[0]; ix.latent_heat = io[1]; ix.water_vapor = io[2];
        ix.x_momentum = io[3]; ix.y_momentum = io[4]; ix.temperature = io[5];
        CHECK(cf_compute_atmosphere_sea_ice_fluxes(ctx, &st, &oc, &e, &ix));
        CHECK(cf_sync(ctx));
        CHECK(cf_d2h(ctx, f, io[5], n * 8)); /* skin temperature [deg C] */
        double skin = NAN;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t k = (size_t)(j + h) * sj + (i + h);
                if (!mask[k]) continue;
                if (isnan(skin)) skin = f[k];
                if (!(f[k] <= 1e-9 && f[k] > -60.0) || fabs(f[k] - skin) > 1e-9) { fprintf(stderr, "sea-ice skin temperature %g (first %g)\n", f[k], skin); return 8; }
            }
    }
    /* error path: invalid parameters are refused with a message */
    cf_flux_params bad = p;
    bad.velocity_difference = 9;
    if (cf_set_flux_params(ctx, &bad) == 0 || !strstr(cf_last_error(ctx), "velocity_formulation")) return 6;
    printf("OK %.6f %.3f\n", first, sum);
    cf_destroy(ctx);
    return 0;
}
