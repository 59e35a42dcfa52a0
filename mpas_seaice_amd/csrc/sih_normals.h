// Device helpers for the normal vectors of cell edges and dual-triangle sides on the sphere
// (/root/reference/src/shared/mpas_seaice_mesh.F:1038-1241, :1393-1606), shared by the weak velocity
// scheme and the upwind advection.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

__device__ inline void rotFwd(double o[3], double x, double y, double z, int rotate)
{  // seaice_grid_rotation_forward, mesh.F:2350-2381
    if (rotate) {
        o[0] = -z;
        o[1] = y;
        o[2] = x;
    } else {
        o[0] = x;
        o[1] = y;
        o[2] = z;
    }
}

// Ry(lat) * Rz(-lon) * in: the rotation onto the rotated equator, mesh.F:1148-1169 and :1199-1201
__device__ inline void toEquator(double o[3], const double in[3], double lat, double lon)
{
    const double cz = cos(-lon), sz = sin(-lon), cy = cos(lat), sy = sin(lat);
    double t[3];
    t[0] = cz * in[0] + (-sz) * in[1] + 0.0 * in[2];
    t[1] = sz * in[0] + cz * in[1] + 0.0 * in[2];
    t[2] = 0.0 * in[0] + 0.0 * in[1] + 1.0 * in[2];
    o[0] = cy * t[0] + 0.0 * t[1] + sy * t[2];
    o[1] = 0.0 * t[0] + 1.0 * t[1] + 0.0 * t[2];
    o[2] = (-sy) * t[0] + 0.0 * t[1] + cy * t[2];
}

// normal of the great circle through a and b, seen in the local east / north directions at mid; mesh.F:1203-1233
__device__ inline void greatCircleNormal(double &n1, double &n2, const double a[3], const double b[3],
                                         const double mid[3], bool flip)
{
    const double vx = b[0] - a[0], vy = b[1] - a[1], vz = b[2] - a[2];
    double g0 = vy * mid[2] - vz * mid[1], g1 = vz * mid[0] - vx * mid[2], g2 = vx * mid[1] - vy * mid[0];
    if (flip) {
        g0 = -1.0 * g0;
        g1 = -1.0 * g1;
        g2 = -1.0 * g2;
    }
    const double norm = sqrt(g0 * g0 + g1 * g1 + g2 * g2);
    g0 = g0 / norm;
    g1 = g1 / norm;
    g2 = g2 / norm;
    double e0 = -mid[1], e1 = mid[0], e2 = 0.0;
    const double en = sqrt(e0 * e0 + e1 * e1);
    e0 = e0 / en;
    e1 = e1 / en;
    e2 = e2 / en;
    n1 = g0 * e0 + g1 * e1 + g2 * e2;
    const double cl = fmax(fmin(n1, 1.0), -1.0);
    n2 = copysign(1.0, g2) * sqrt(1.0 - cl * cl);
}

