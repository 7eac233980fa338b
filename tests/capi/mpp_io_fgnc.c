/* mpp_io_fgnc.c -- TEST HARNESS ONLY: the two mpp_io entry points integration/field_io_hip.c calls
 * (mpp_get_var_value_block / mpp_put_var_value_block, tools/libfrencutils/mpp_io.h:53,76) served by OUR classic-netCDF
 * reader / writer (fg_nc_*, csrc/field_file.c), because this image has no libnetcdf to build the reference's mpp_io.c with.
 * Same type dispatch as mpp_io.c:443-470 / :1349-1380: NC_DOUBLE and NC_FLOAT variables travel as double, NC_INT / NC_SHORT
 * in their own type.  Not a build of the reference and not an oracle: it lets tests/capi/field_io_driver.c run OUR
 * replacement objects the way fregrid.c runs the originals. */
#include <stdio.h>
#include <stdlib.h>
#include "mpp.h"
#include "mpp_io.h"
#include "fregrid_hip.h"

#define SHIM_MAXFILES 64
static fg_ncfile *g_files[SHIM_MAXFILES];
static int g_nfiles = 0;

int shim_register_file(fg_ncfile *f)
{
  if (g_nfiles >= SHIM_MAXFILES) mpp_error("mpp_io shim: too many files");
  g_files[g_nfiles] = f;
  return g_nfiles++;
}
static int var_type(int fid, int vid, int *ndims)
{
  int type = 0;
  if (fid < 0 || fid >= g_nfiles) mpp_error("mpp_io shim: invalid fid");
  if (fg_nc_inq_var(g_files[fid], vid, NULL, 0, &type, ndims, NULL, NULL)) mpp_error((char *)fg_nc_last_error());
  return type;
}
void mpp_get_var_value_block(int fid, int vid, const size_t *start, const size_t *nread, void *data)
{
  long st[8], cn[8];
  int nd = 0, k, rc;
  const int type = var_type(fid, vid, &nd);
  for (k = 0; k < nd; k++) { st[k] = (long)start[k]; cn[k] = (long)nread[k]; }
  if (type == FG_NC_DOUBLE || type == FG_NC_FLOAT) rc = fg_nc_get_vara_double(g_files[fid], vid, st, cn, (double *)data);
  else rc = fg_nc_get_vara(g_files[fid], vid, st, cn, data);
  if (rc) mpp_error((char *)fg_nc_last_error());
}
void mpp_put_var_value_block(int fid, int vid, const size_t *start, const size_t *nwrite, const void *data)
{
  long st[8], cn[8];
  int nd = 0, k, rc;
  const int type = var_type(fid, vid, &nd);
  if (mpp_pe() != mpp_root_pe()) return;
  for (k = 0; k < nd; k++) { st[k] = (long)start[k]; cn[k] = (long)nwrite[k]; }
  if (type == FG_NC_DOUBLE || type == FG_NC_FLOAT) rc = fg_nc_put_vara_double(g_files[fid], vid, st, cn, (const double *)data);
  else rc = fg_nc_put_vara(g_files[fid], vid, st, cn, data);
  if (rc) mpp_error((char *)fg_nc_last_error());
}
