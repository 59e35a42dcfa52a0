"""Analytic known-answer fields restated from the reference's idealised test cases.

Sphere: real spherical harmonics u = Y_5^3, v = Y_4^2 on the rotated grid and their strains
(/root/reference/testing_and_setup/testcases/spherical_operators/strain/create_ic.py:470-588).
Plane: linear and sinusoid velocity fields with their strains and stress divergence
(/root/reference/testing_and_setup/testcases/square/operators_strain/create_ics.py:7-116).
The reference scripts cannot be imported here (they need netCDF4, which is absent), so the
formulas are restated and pinned to the sample values the reference's functions returned when
they were run during the survey of the reference.
"""
import numpy as np
from math import factorial, pi
from scipy.special import lpmv


def _K(m, l):
    return np.sqrt(((2 * l + 1) * factorial(l - abs(m))) / (4.0 * pi * factorial(l + abs(m))))


def _P(m, l, x):
    # the reference tabulates associated Legendre functions without the Condon-Shortley phase
    return (-1.0) ** m * lpmv(m, l, x)


def _dP(m, l, x):
    return (l * x * _P(m, l, x) - (l + m) * _P(m, l - 1, x)) / (x * x - 1.0)


def real_harmonic(lon, lat, m, l):
    x = np.sin(lat)  # cos(theta), theta = pi/2 - lat
    assert m > 0
    return np.sqrt(2.0) * _K(m, l) * np.cos(m * lon) * _P(m, l, x)


def real_harmonic_derivs(lon, lat, m, l):
    x = np.sin(lat)
    # d/dlat P(sin lat) = P'(x) cos(lat)
    dlat = np.sqrt(2.0) * _K(m, l) * np.cos(m * lon) * _dP(m, l, x) * np.cos(lat)
    dlon = -m * np.sqrt(2.0) * _K(m, l) * np.sin(m * lon) * _P(m, l, x)
    return dlon, dlat


def velocities_strains_analytical(lat, lon, mu=3, lu=5, mv=2, lv=4, r=1.0):
    u = real_harmonic(lon, lat, mu, lu)
    v = real_harmonic(lon, lat, mv, lv)
    du_dlon, du_dlat = real_harmonic_derivs(lon, lat, mu, lu)
    dv_dlon, dv_dlat = real_harmonic_derivs(lon, lat, mv, lv)
    e11 = (1.0 / (r * np.cos(lat))) * (du_dlon - v * np.sin(lat))
    e22 = dv_dlat / r
    e12 = (0.5 / r) * (du_dlat + u * np.tan(lat) + dv_dlon / np.cos(lat))
    return u, v, e11, e22, e12


def _d2P(m, l, x):
    # Legendre's equation: (1 - x^2) P'' - 2 x P' + (l (l + 1) - m^2 / (1 - x^2)) P = 0
    return (2.0 * x * _dP(m, l, x) - (l * (l + 1) - m * m / (1.0 - x * x)) * _P(m, l, x)) / (1.0 - x * x)


def real_harmonic_second_derivs(lon, lat, m, l):
    """d2/dlon2, d2/dlondlat, d2/dlat2 of the real harmonic."""
    x, c = np.sin(lat), np.cos(lat)
    k = np.sqrt(2.0) * _K(m, l)
    lonlon = -m * m * k * np.cos(m * lon) * _P(m, l, x)
    lonlat = -m * k * np.sin(m * lon) * _dP(m, l, x) * c
    latlat = k * np.cos(m * lon) * (_d2P(m, l, x) * c * c - _dP(m, l, x) * x)
    return lonlon, lonlat, latlat


def stress_divergence_analytical(lat, lon, mu=3, lu=5, mv=2, lv=4, r=1.0):
    """divergence of the stress field sigma = strain(u, v) on the sphere, metric terms included
    (/root/reference/testing_and_setup/testcases/spherical_operators/stress_divergence/create_ic.py:574-603)"""
    u, v, e11, e22, e12 = velocities_strains_analytical(lat, lon, mu, lu, mv, lv, r)
    du_dlon, du_dlat = real_harmonic_derivs(lon, lat, mu, lu)
    dv_dlon, dv_dlat = real_harmonic_derivs(lon, lat, mv, lv)
    d2u_lonlon, d2u_lonlat, d2u_latlat = real_harmonic_second_derivs(lon, lat, mu, lu)
    d2v_lonlon, d2v_lonlat, d2v_latlat = real_harmonic_second_derivs(lon, lat, mv, lv)
    c, t = np.cos(lat), np.tan(lat)
    de11_dlon = (1.0 / (r * c)) * (d2u_lonlon - dv_dlon * np.sin(lat))
    de22_dlat = d2v_latlat / r
    de12_dlon = (0.5 / r) * (d2u_lonlat + du_dlon * t + d2v_lonlon / c)
    de12_dlat = (0.5 / r) * (d2u_latlat + du_dlat * t + u / c ** 2 + d2v_lonlat / c + dv_dlon * (t / c))
    divu = (1.0 / (r * c)) * de11_dlon + (1.0 / r) * de12_dlat - (2.0 / r) * t * e12
    divv = (1.0 / (r * c)) * de12_dlon + (1.0 / r) * de22_dlat + (1.0 / r) * t * e11 - (1.0 / r) * t * e22
    return divu, divv


# values of the reference's velocities_strains_analytical(0.7, 1.1, 3, 5, 2, 4)
SPHERE_SAMPLE = dict(lat=0.7, lon=1.1, u=-0.59121435168, v=-0.31027979175, e11=0.63179002647,
                     e22=-0.60078848715, e12=-1.01789907063)


def planar_fields(x, y, kind="sinusoid", L=1.0, A=2.56):
    """u, v, e11, e22, e12, divu, divv for the planar operator cases (domain side L, wavenumber A)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if kind == "linear":
        u, v = x - 0.5 * L, y - 0.5 * L
        one, zero = np.ones_like(x), np.zeros_like(x)
        return u, v, one, one, zero, zero, zero
    k = 2.0 * pi * A / L
    sx, cx, sy, cy = np.sin(k * x), np.cos(k * x), np.sin(k * y), np.cos(k * y)
    u = sx * sy
    v = sx * sy
    dudx, dudy = k * cx * sy, k * sx * cy
    dvdx, dvdy = k * cx * sy, k * sx * cy
    d2udx2 = d2udy2 = d2vdx2 = d2vdy2 = -k * k * sx * sy
    d2udxdy = d2vdxdy = k * k * cx * cy
    e11, e22, e12 = dudx, dvdy, 0.5 * (dudy + dvdx)
    divu = d2udx2 + 0.5 * (d2udy2 + d2vdxdy)
    divv = 0.5 * (d2udxdy + d2vdx2) + d2vdy2
    return u, v, e11, e22, e12, divu, divv


# values of the reference's velocities_strains_stress_divergences(0.3, 0.4, "sinusoid")
PLANE_SAMPLE = dict(x=0.3, y=0.4, u=-0.14926584448, v=-0.14926584448, e11=0.27270298007,
                    e22=-15.8008226061, e12=-7.76405981304, divu=72.3621268644, divv=72.3621268644)
