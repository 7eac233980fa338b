/* SHIM, not a netCDF header.  This image has no libnetcdf; tools/libfrencutils/globals.h includes <netcdf.h> only for the
 * `nc_type` typedef (an int in netCDF) of struct members the conservative-interpolation path never reads.  This directory goes on
 * the include path in two places, both of which compile OUR code against the reference's own struct and prototype declarations
 * (globals.h, conserve_interp.h, mpp.h) -- never the reference itself, and never anything used as an oracle:
 *   tests/test_capi_c.py      gcc -fsyntax-only of integration/conserve_interp_hip.c
 *   oracle/Makefile           oracle/_ref/b2_driver = tests/capi/b2_driver.c + integration/conserve_interp_hip.c
 *                             (+ the reference's mpp.c / mpp_domain.c, which need no netCDF), run by tests/test_gpu_b2_driver.py */
#ifndef FG_TYPECHECK_NETCDF_SHIM
#define FG_TYPECHECK_NETCDF_SHIM
typedef int nc_type;
#endif
