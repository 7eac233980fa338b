/*
 * fregrid_hip_glue.h -- hand-over of device-resident fields between the two replacement objects of this directory:
 * integration/field_io_hip.c (get_input_data / write_field_data) and integration/conserve_interp_hip.c
 * (do_scalar_conserve_interp).  Both symbols are referenced WEAKLY by conserve_interp_hip.c, so that object still links and
 * runs alone (then it stages the host arrays of Field_config itself, as before).
 */
#ifndef FREGRID_HIP_GLUE_H_
#define FREGRID_HIP_GLUE_H_
#include "globals.h"

typedef struct {
  double *d_data;      /* [nz][f_stride]: tiles back to back, halo'd for conserve_order2 (halo filled), scaled / offset          */
  double *d_gx, *d_gy; /* [nz][ncell] gradients (order 2), or NULL                                                            */
  int *d_gm;           /* [ncell] gradient mask (order 2 with missing values), or NULL                                          */
  int nz, order, varid;
  long f_stride, ncell;
} FgDevField;

/* the device copy get_input_data made for this field array (and variable), or NULL */
const FgDevField *fg_glue_input(const Field_config *field, int varid);
/* do_scalar_conserve_interp hands the remapped levels of output tile `field_out_n` over (takes ownership of d_out [nz][n]);
 * write_field_data narrows and downloads them in the file's type.  Returns 0 when nobody wants them (the caller frees). */
int fg_glue_output_put(const Field_config *field_out_n, double *d_out, long n, int nz);
#endif
