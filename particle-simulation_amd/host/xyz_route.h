/* The x layer of a point, for routing points to slabs on the host: the reference's cell
 * formula (main.c:134, (int)((f - (-10.)) / 0.05) with f promoted to double), or -1 for a
 * point outside the 400 layers.  Evaluated in binary32 first: u = fl(fl(f + 10) * 20)
 * differs from the reference's quotient by less than 3.5e-5 on [0, 401) (half an ulp of
 * f + 10 <= 2^-20, times 20, plus half an ulp of the product <= 2^-16), so at least 1e-3
 * away from an integer floor(u) is the reference's index; closer to a cell boundary, below
 * zero and beyond the grid the reference's own expression decides.
 * xyz_selfcheck_route_layers (xyzhost.c) compares the two over any range of bit patterns. */
#ifndef XYZ_ROUTE_H
#define XYZ_ROUTE_H

#define XYZ_ROUTE_LAYERS 400

static inline int xyz_route_layer_exact(float f)
{
    const double q = ((double)f + 10.0) / 0.05;
    return q >= 0.0 && q < (double)XYZ_ROUTE_LAYERS ? (int)q : -1;
}

static inline int xyz_route_layer(float f)
{
    const float u = (f + 10.0f) * 20.0f;
    const int c = (int)u;                                  /* (u in [0, 400.5) below: no overflow matters) */
    const float fr = u - (float)c;
    if (!(fr >= 1.0e-3f && fr <= 0.999f && u >= 0.0f && u < 400.5f)) return xyz_route_layer_exact(f);
    return c < XYZ_ROUTE_LAYERS ? c : -1;
}

#endif
