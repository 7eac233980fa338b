/* SHIM, not a netCDF header.  This image has no libnetcdf; tools/libfrencutils/globals.h includes <netcdf.h> only for the
 * `nc_type` typedef (an int in netCDF) of struct members the conservative-interpolation path never reads.  This directory goes on
 * the include path in two places, both of which compile OUR code against the reference's own struct and prototype declarations
 * (globals.h, conserve_interp.h, mpp.h) -- never the reference itself, and never anything used as an oracle:
 *   tests/test_capi_c.py      gcc -fsyntax-only of integration/conserve_interp_hip.c and integration/field_io_hip.c
 *   oracle/Makefile           oracle/_ref/b2_driver = tests/capi/b2_driver.c + integration/conserve_interp_hip.c
 *                             (+ the reference's mpp.c / mpp_domain.c, which need no netCDF), run by tests/test_gpu_b2_driver.py;
 *                             oracle/_ref/field_io_driver likewise (tests/capi/field_io_driver.c + the two integration objects) */
#ifndef FG_TYPECHECK_NETCDF_SHIM
#define FG_TYPECHECK_NETCDF_SHIM
typedef int nc_type;
/* the external type codes of netCDF's nc_type (public, fixed by the file format: 1 byte, 2 char, 3 short, 4 int, 5 float, 6 double):
 * integration/field_io_hip.c switches on Var_config.type as the reference's get_input_data / write_field_data do */
#define NC_BYTE 1
#define NC_CHAR 2
#define NC_SHORT 3
#define NC_INT 4
#define NC_FLOAT 5
#define NC_DOUBLE 6
#endif
