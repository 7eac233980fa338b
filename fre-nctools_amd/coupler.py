"""Atmosphere x land x ocean exchange grid of make_coupler_mosaic (tools/make_coupler_mosaic/make_coupler_mosaic.c:1360-1727,
legacy clip, one tile per component grid or several atmosphere tiles) as THREE searches on the device instead of three nested
brute-force loops:

  1. plan(atm -> lnd): the atmosphere x land cells and their clipped polygons (fg_plan_get_polygons) -- the reference keeps the
     vertices of each (atmxlnd_x / _y, :1560-1577);
  2. plan(atm -> ocn): the atmosphere x ocean cells, weighted by the ocean fraction (:1604-1657);
  3. plan(polygon list -> ocn) (fg_plan_create_polylist): every atmosphere x land polygon clipped against the ocean cells it
     overlaps, weighted by the land fraction and added up per polygon in ocean-cell order (:1659-1692), then the final area
     test of the atmosphere x land cell (:1700-1716).

The plans apply create_xgrid's acceptance (area / min(area_src, area_dst) > 1e-6, no weights); the coupler's own tests
(area x fraction against its own reference areas) can only reject more, so they are applied here, on the host, to the plans'
lists -- same products, same order, same bits as the reference loop (tests/test_gpu_coupler.py drives the reference's own
compiled clip_2dx2d / poly_area / poly_ctrlon / fix_lon through that loop and compares bit for bit).

Host logic only; all geometry runs in libfregrid_hip.  Not covered: nested atmosphere tiles, the wave grid, great-circle clip,
lnd_same_as_atm's vertex copy (pass distinct grids), the land x ocean list and the mask / centroid bookkeeping after :1727.
"""
import numpy as np

from .conserve_interp import GridConfig, XgridPlan

MIN_AREA_FRAC = 1.0e-4          # make_coupler_mosaic.c:148
AREA_RATIO_THRESH = 1.0e-6      # :362


def coupler_xgrid(atm, lnd, ocn, omask, interp_order=2, area_ratio_thresh=AREA_RATIO_THRESH, device=0):
    """atm: list of GridConfig (atmosphere tiles), lnd / ocn: one GridConfig each, omask [ny_ocn, nx_ocn] ocean fraction.
    Returns dict(axo=..., axl=...): axo = atmosphere x ocean cells (ta, ia, ja, io, jo, area[, clon, clat]) in the reference's
    order (atmosphere cell major, ocean cell ascending); axl = atmosphere x land cells (ta, ia, ja, il, jl, area[, clon, clat])."""
    from . import get_grid_area
    from ._lib import lib
    lib().fg_set_search_frame(1)            # the coupler's longitude convention for the second cell of a pair (fix_lon(cell, xa_avg))
    try:
        return _coupler_xgrid(atm, lnd, ocn, omask, interp_order, area_ratio_thresh, device, get_grid_area)
    finally:
        lib().fg_set_search_frame(0)


def _coupler_xgrid(atm, lnd, ocn, omask, interp_order, area_ratio_thresh, device, get_grid_area):
    order = 2 if interp_order == 2 else 1
    omask = np.ascontiguousarray(omask, dtype=np.float64).reshape(-1)
    area_atm = np.concatenate([get_grid_area(g.nx, g.ny, g.lonc, g.latc) for g in atm])
    area_lnd = get_grid_area(lnd.nx, lnd.ny, lnd.lonc, lnd.latc)
    area_ocn = get_grid_area(ocn.nx, ocn.ny, ocn.lonc, ocn.latc)
    off = np.concatenate([[0], np.cumsum([g.nx * g.ny for g in atm])])
    nxa = np.array([g.nx for g in atm])

    def atm_cell(x):
        return off[x["t_in"]] + x["j_in"].astype(np.int64) * nxa[x["t_in"]] + x["i_in"]

    # --- 1. atmosphere x land candidates and their polygons (:1398-1585)
    p = XgridPlan.create(order, atm, lnd, device=device)
    xl = p.get_xgrid()
    poly = p.get_polygons(maxv=8)
    src_struct = p.get_cell_struct(0, int(off[-1]))
    p.destroy()
    xa_avg = src_struct["lon_avg"]                          # avgval_double(na_in, xa) after fix_lon(xa, ya, 4, M_PI), :1389-1394
    a_of_l = atm_cell(xl)
    lcell = xl["j_out"].astype(np.int64) * lnd.nx + xl["i_out"]
    if np.any(poly["n"] > 8):
        raise ValueError("coupler_xgrid: an atmosphere x land polygon has more than 8 vertices")
    min_area_l = np.minimum(area_lnd[lcell], area_atm[a_of_l])          # :1701

    # --- 2. atmosphere x ocean over sea (:1604-1657)
    p = XgridPlan.create(order, atm, ocn, device=device)
    xo = p.get_xgrid()
    p.destroy()
    ocell = xo["j_out"].astype(np.int64) * ocn.nx + xo["i_out"]
    frac = omask[ocell]
    xarea = xo["area"] * frac                                           # poly_area(...) * ocn_frac, :1634
    keep = (frac > MIN_AREA_FRAC) & (xarea / np.minimum(area_ocn[ocell], area_atm[atm_cell(xo)]) > area_ratio_thresh)
    axo = {"ta": xo["t_in"][keep], "ia": xo["i_in"][keep], "ja": xo["j_in"][keep], "io": xo["i_out"][keep], "jo": xo["j_out"][keep],
           "area": xarea[keep]}
    if order == 2:
        axo["clon"], axo["clat"] = xo["c1"][keep] * frac[keep], xo["c2"][keep] * frac[keep]      # :1648-1649

    # --- 3. the remembered atmosphere x land polygons against the ocean cells over land (:1659-1692)
    axl_area = np.zeros(poly["n"].size)
    axl_clon, axl_clat = np.zeros_like(axl_area), np.zeros_like(axl_area)
    if poly["n"].size:
        p = XgridPlan.create_polylist(order, poly["n"], poly["lon"], poly["lat"], xa_avg[a_of_l], min_area_l, ocn, device=device)
        xp = p.get_xgrid()
        p.destroy()
        l = xp["i_in"].astype(np.int64)
        oc = xp["j_out"].astype(np.int64) * ocn.nx + xp["i_out"]
        lfrac = 1.0 - omask[oc]
        xa = xp["area"] * lfrac                                         # :1673
        ok = (lfrac > MIN_AREA_FRAC) & (xa / min_area_l[l] > area_ratio_thresh)
        # axl_area[l] += xarea in ocean-cell order: the plan's order for one polygon is ascending ocean cell, and np.add.at adds
        # the selected terms one by one in array order
        np.add.at(axl_area, l[ok], xa[ok])
        if order == 2:
            np.add.at(axl_clon, l[ok], xp["c1"][ok] * lfrac[ok])
            np.add.at(axl_clat, l[ok], xp["c2"][ok] * lfrac[ok])
    fin = axl_area / min_area_l > area_ratio_thresh                     # :1704
    axl = {"ta": xl["t_in"][fin], "ia": xl["i_in"][fin], "ja": xl["j_in"][fin], "il": xl["i_out"][fin], "jl": xl["j_out"][fin],
           "area": axl_area[fin]}
    if order == 2:
        axl["clon"], axl["clat"] = axl_clon[fin], axl_clat[fin]
    return {"axo": axo, "axl": axl, "n_axl_polygons": int(poly["n"].size)}
