/* TYPE-CHECK SHIM, not a netCDF header.  This image has no libnetcdf; tools/libfrencutils/globals.h includes <netcdf.h> only for
 * the nc_type typedef of three struct members.  tests/test_capi_c.py puts this directory on the include path for ONE purpose:
 * to let gcc -fsyntax-only check integration/conserve_interp_hip.c against the reference's own struct and prototype
 * declarations (globals.h, conserve_interp.h, mpp.h).  Nothing compiled with it is linked, run, or used as an oracle. */
#ifndef FG_TYPECHECK_NETCDF_SHIM
#define FG_TYPECHECK_NETCDF_SHIM
typedef int nc_type;
#endif
