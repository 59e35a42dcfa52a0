// Mechanical redistribution (ridging) of the column package on the device: ridge_ice of
// src/column/ice_mechred.F90:80-597 with asum_ridging :609, ridge_prep :640, ridge_itd :741, ridge_shift :1030 and
// colpkg_compute_tracers (src/column/ice_colpkg_tracers.F90:163-254), run for every owned cell as column_ridging does
// (src/shared/mpas_seaice_column.F:3468-3530 -> colpkg_step_ridge, src/column/ice_colpkg.F90:3296-3325).
//
// One thread per owned cell.  The category arrays of a cell (areas, volumes, participation, ridge thicknesses) live
// in the thread; the tracers stay where the advection left them, in the planes of the device fields (plane =
// category * layers + layer, the cell index fastest), so a wave reads and writes every plane as one coalesced
// stream, and nothing is gathered into per-cell records.  The products area*tracer / volume*tracer that ridge_shift
// keeps while it moves ice between categories (atrcrn, :1151) are a scratch field of the same shape.
//
// Scope: tracers without strata (n_trcr_strata = 0): every advected tracer other than the ice area and the two
// volumes must have the ice area, the ice volume or the snow volume as its parent -- the standard physics set
// (surface temperature, ice enthalpy, ice salinity, snow enthalpy).  Tracers carried on other tracers (ponds, brine,
// level ice, aerosols: the strata branches :1206-1211, :1389-1394, :1508-1517) are refused.  The module's
// l_conservation_check is .false. (:64).  The ITD clean-up that colpkg_step_ridge runs after ridge_ice (cleanup_itd,
// src/column/ice_itd.F90:754-1020, over the thermodynamic step dt * ndtd, src/column/ice_colpkg.F90:3327-3354) follows
// in the same thread: rebin with shift_ice, zap_small_areas, zap_snow_temperature, for the same tracer set.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sih_internal.h"

#define RIDGE_MAXCAT 10
#define RIDGE_MAXTR 48
#define RIDGE_BLOCK 64  // cells (threads) per block of the ridging kernel
typedef __attribute__((address_space(3))) double RidgeLdsDouble;

struct RidgeTracer {
    double *val;  // plane 0 of the field
    int L, layer, depend, pad;
};

struct RidgeDev {
    int nCellsSolve, padC, ncat, ntrcr, nslyr, ntTsfc, ntQsno, krdgPartic, krdgRedist, ndtd;
    int cleanup, nilyr, ntQice;  // the ITD clean-up after the ridging; ice layers and the first ice-enthalpy tracer
    double *aice, *fsalt;        // iceAreaCell, oceanSaltFlux
    int *firstIce;               // newlyFormedIce (nCategories planes)
    double dt, muRdg;
    double hinMax[RIDGE_MAXCAT + 1];
    double *aicen, *vicen, *vsnon, *aice0, *atrcrn;
    const double *rdgConv, *rdgShear, *Tf;
    double *dardg1dt, *dardg2dt, *dvirdgdt, *opening, *fresh, *fhocn;
    double *aparticn, *krdgn, *aredistn, *vredistn, *dardg1ndt, *dardg2ndt, *dvirdgndt, *araftn, *vraftn;
    int *flag;  // [0] highest stop code of any cell, [1] most iterations of any cell
    // the tracers, in device memory (indexed by run-time values: as a member array the whole argument block would be
    // copied to scratch memory for it)
    const RidgeTracer *tr;
};

struct sih_ridge_state {
    double *atrcrn = nullptr;
    int planes = 0;
    int *d_flag = nullptr;
    int lastIterations = 0;
    RidgeTracer *d_tr = nullptr;       // RIDGE_MAXTR entries
    std::vector<RidgeTracer> hostTr;   // what d_tr holds
};

void sih_ridge_destroy(sih_ctx *c)
{
    if (!c->ridge) return;
    if (c->ridge->d_flag) (void)hipFree(c->ridge->d_flag);
    if (c->ridge->d_tr) (void)hipFree(c->ridge->d_tr);
    delete c->ridge;
    c->ridge = nullptr;
}

#define PUNY 1.0e-11
// ice_mechred.F90:51-61, ice_constants_colpkg.F90:24
#define RDG_CS 0.25
#define RDG_FSNOW 0.5
#define RDG_GSTAR 0.15
#define RDG_ASTAR 0.05
#define RDG_MAXRAFT 1.0
#define RDG_HSTAR 25.0
#define RDG_RHOS 330.0

// Loops over the categories run to the compile-time bound, so that every index into the per-cell arrays is a constant
// and the arrays stay in registers (with run-time bounds they lived in scratch memory)
#define FORCAT(n) _Pragma("unroll") for (int n = 1; n <= NC; n++) if (n <= ncat)
#define FORCAT0(n) _Pragma("unroll") for (int n = 0; n <= NC; n++) if (n <= ncat)

// category arrays are 1-based (index 0 of apartic is the open water), as in the reference
template <int NC>
struct RidgeCell {
    double aicen[NC + 1], vicen[NC + 1], vsnon[NC + 1];
    double apartic[NC + 1], hrmin[NC + 1], hrmax[NC + 1], hrexp[NC + 1], krdg[NC + 1];
    double ardg1n[NC + 1], ardg2n[NC + 1], virdgn[NC + 1], mraftn[NC + 1];
};

// ridge_itd :741-1012
template <int NC>
__device__ __forceinline__ void ridgeItd(const RidgeDev &d, int ncat, double aice0, RidgeCell<NC> &s, double &aksum)
{
    const double Gstari = 1.0 / RDG_GSTAR, astari = 1.0 / RDG_ASTAR;
    double GsumA[NC + 2];
    double *Gsum = GsumA + 1;
    Gsum[-1] = 0.0;
    Gsum[0] = aice0 > PUNY ? aice0 : Gsum[-1];
    s.apartic[0] = 0.0;
    FORCAT(n) {
        Gsum[n] = 1.0;
        s.apartic[n] = 0.0;
        s.hrmin[n] = 0.0;
        s.hrmax[n] = 0.0;
        s.hrexp[n] = 0.0;
        s.krdg[n] = 1.0;
        if (s.aicen[n] > PUNY)
            Gsum[n] = Gsum[n - 1] + s.aicen[n];
        else
            Gsum[n] = Gsum[n - 1];
    }
    const double work = 1.0 / Gsum[ncat];
    FORCAT0(n) Gsum[n] = Gsum[n] * work;
    if (d.krdgPartic == 0) {
        FORCAT0(n) {
            if (Gsum[n] < RDG_GSTAR)
                s.apartic[n] = Gstari * (Gsum[n] - Gsum[n - 1]) * (2.0 - Gstari * (Gsum[n - 1] + Gsum[n]));
            else if (Gsum[n - 1] < RDG_GSTAR)
                s.apartic[n] = Gstari * (RDG_GSTAR - Gsum[n - 1]) * (2.0 - Gstari * (RDG_GSTAR + Gsum[n - 1]));
        }
    } else {
        const double xtmp = 1.0 / (1.0 - exp(-astari));
        Gsum[-1] = exp(-Gsum[-1] * astari) * xtmp;
        FORCAT0(n) {
            Gsum[n] = exp(-Gsum[n] * astari) * xtmp;
            s.apartic[n] = Gsum[n - 1] - Gsum[n];
        }
    }
    if (d.krdgRedist == 0) {
        FORCAT(n)
            if (s.aicen[n] > PUNY) {
                const double hi = s.vicen[n] / s.aicen[n];
                s.hrmin[n] = fmin(2.0 * hi, hi + RDG_MAXRAFT);
                s.hrmax[n] = 2.0 * sqrt(RDG_HSTAR * hi);
                s.hrmax[n] = fmax(s.hrmax[n], s.hrmin[n] + PUNY);
                const double hrmean = 0.5 * (s.hrmin[n] + s.hrmax[n]);
                s.krdg[n] = hrmean / hi;
            }
    } else {
        FORCAT(n)
            if (s.aicen[n] > PUNY) {
                double hi = s.vicen[n] / s.aicen[n];
                hi = fmax(hi, PUNY);
                s.hrmin[n] = fmin(2.0 * hi, hi + RDG_MAXRAFT);
                s.hrexp[n] = d.muRdg * sqrt(hi);
                s.krdg[n] = (s.hrmin[n] + s.hrexp[n]) / hi;
                s.mraftn[n] = fmax(0.0, copysign(1.0, hi + RDG_MAXRAFT - s.hrmin[n]));  // :976-980
                const double xtmp = s.mraftn[n] * ((2.0 * hi + s.hrexp[n]) / hi - s.krdg[n]);
                s.mraftn[n] = fmax(0.0, copysign(1.0, PUNY - fabs(xtmp)));
            }
    }
    aksum = s.apartic[0];
    FORCAT(n) aksum = aksum + s.apartic[n] * (1.0 - 1.0 / s.krdg[n]);
}

__device__ inline double &trc(const RidgeDev &d, int it, int n, int cell)  // trcrn(it, n), n = 1 .. ncat
{
    const RidgeTracer &t = d.tr[it];
    return t.val[((size_t)(n - 1) * t.L + t.layer) * d.padC + cell];
}
__device__ inline double &atr(const RidgeDev &d, int it, int n, int cell)  // atrcrn(it, n)
{
    return d.atrcrn[((size_t)it * d.ncat + (n - 1)) * d.padC + cell];
}

// ridge_shift :1030-1541; returns the number of the reference's stop (1 'aice0 < 0', 2 'ardg > aicen') or 0.
// The reference moves the tracer products of a ridging category to the receiving ones inside its loops over category
// pairs (:1206-1211, :1508-1517).  Here those loops only keep what the products need (the areas and volumes that
// leave category n, the fractions farea / fvol that category nr receives of them), and the products of one tracer
// at a time are then formed, moved and divided again in registers, in the reference's order of operations for every
// product: a tracer is read once and written once, and the tracers are independent of each other.
template <int NC>
__device__ __forceinline__ int ridgeShift(const RidgeDev &d, int ncat, int cell, RidgeCell<NC> &s, double &aice0, double aksum,
                          double closing_net, double &opning, double &ardg1, double &ardg2, double &virdg, double &aopen,
                          double &msnow_mlt, double &esnow_mlt, double Tf, RidgeLdsDouble *frac)
{
    double aicen_init[NC + 1], vicen_init[NC + 1], vsnon_init[NC + 1];
    const int ntrcr = d.ntrcr;
    const double dt = d.dt;
    FORCAT(n)
    {
        aicen_init[n] = s.aicen[n];
        vicen_init[n] = s.vicen[n];
        vsnon_init[n] = s.vsnon[n];
    }
    double closing_gross = closing_net / aksum;
    if (s.apartic[0] > 0.0) {
        const double wk1 = s.apartic[0] * closing_gross * dt;
        if (wk1 > aice0) {
            const double tmpfac = aice0 / wk1;
            closing_gross = closing_gross * tmpfac;
            opning = opning * tmpfac;
        }
    }
    FORCAT(n)
    if (s.aicen[n] > PUNY && s.apartic[n] > 0.0) {
        const double wk1 = s.apartic[n] * closing_gross * dt;
        if (wk1 > s.aicen[n]) {
            const double tmpfac = s.aicen[n] / wk1;
            closing_gross = closing_gross * tmpfac;
            opning = opning * tmpfac;
        }
    }
    aice0 = aice0 - s.apartic[0] * closing_gross * dt + opning * dt;
    if (aice0 < -PUNY)
        return 1;
    else if (aice0 < 0.0)
        aice0 = 0.0;
    aopen = opning * dt;

    // category n ridges in this call: ardg1n, ardg2n, virdgn of the cell's state and vsrdgnC are those of this call.
    // The fractions farea(n, nr), fvol(n, nr) wait for the tracers in LDS (`frac`: this thread's column of
    // [2 * NC * NC][RIDGE_BLOCK]): 50 more doubles would not fit the registers beside the cell's state
    bool moves[NC + 1];
    double vsrdgnC[NC + 1];
    bool any = false;
    FORCAT(n)
    {
        moves[n] = false;
        if (aicen_init[n] > PUNY && s.apartic[n] > 0.0 && closing_gross > 0.0) {
            double ardg1n = s.apartic[n] * closing_gross * dt;
            if (ardg1n > aicen_init[n] + PUNY)
                return 2;
            else
                ardg1n = fmin(aicen_init[n], ardg1n);
            const double ardg2n = ardg1n / s.krdg[n];
            const double afrac = ardg1n / aicen_init[n];
            const double virdgn = vicen_init[n] * afrac;
            const double vsrdgn = vsnon_init[n] * afrac;
            s.aicen[n] = s.aicen[n] - ardg1n;
            s.vicen[n] = s.vicen[n] - virdgn;
            s.vsnon[n] = s.vsnon[n] - vsrdgn;
            ardg1 = ardg1 + ardg1n;
            ardg2 = ardg2 + ardg2n;
            virdg = virdg + virdgn;
            s.ardg1n[n] = ardg1n;
            s.ardg2n[n] = ardg2n;
            s.virdgn[n] = virdgn;
            moves[n] = true;
            any = true;
            vsrdgnC[n] = vsrdgn;
            msnow_mlt = msnow_mlt + RDG_RHOS * vsrdgn * (1.0 - RDG_FSNOW);
            const double dhr = s.hrmax[n] - s.hrmin[n];
            const double dhr2 = s.hrmax[n] * s.hrmax[n] - s.hrmin[n] * s.hrmin[n];
            if (d.ntQsno >= 0)
                for (int k = 1; k <= d.nslyr; k++) {
                    const double esrdgn = vsrdgn * trc(d, d.ntQsno + k - 1, n, cell) / (double)d.nslyr;
                    esnow_mlt = esnow_mlt + esrdgn * (1.0 - RDG_FSNOW);
                }
            double lastExpR = 0.0;
            bool haveExp = false;
            FORCAT(nr)
            {
                double farea, fvol, hL, hR;
                if (d.krdgRedist == 0) {
                    if (s.hrmin[n] >= d.hinMax[nr] || s.hrmax[n] <= d.hinMax[nr - 1]) {
                        hL = 0.0;
                        hR = 0.0;
                    } else {
                        hL = fmax(s.hrmin[n], d.hinMax[nr - 1]);
                        hR = fmin(s.hrmax[n], d.hinMax[nr]);
                    }
                    farea = (hR - hL) / dhr;
                    fvol = (hR * hR - hL * hL) / dhr2;
                } else {
                    // expL of this category is expR of the one before whenever that one received ice: its hR was
                    // hinMax[nr - 1] > hi1, which is this hL, so the same expression is not evaluated twice
                    const double hi1 = s.hrmin[n], hexp = s.hrexp[n];
                    if (nr < ncat) {
                        if (hi1 >= d.hinMax[nr]) {
                            farea = 0.0;
                            fvol = 0.0;
                            haveExp = false;
                        } else {
                            hL = fmax(hi1, d.hinMax[nr - 1]);
                            hR = d.hinMax[nr];
                            const double expL = haveExp ? lastExpR : exp(-(hL - hi1) / hexp);
                            const double expR = exp(-(hR - hi1) / hexp);
                            farea = expL - expR;
                            fvol = ((hL + hexp) * expL - (hR + hexp) * expR) / (hi1 + hexp);
                            lastExpR = expR;
                            haveExp = true;
                        }
                    } else {
                        hL = fmax(hi1, d.hinMax[nr - 1]);
                        const double expL = haveExp ? lastExpR : exp(-(hL - hi1) / hexp);
                        farea = expL;
                        fvol = (hL + hexp) * expL / (hi1 + hexp);
                    }
                    if (n == 1) {  // diagnostics of the thinnest ridging ice :1462-1469
                        // (intent(inout) in the reference: the field keeps its value where nothing ridges)
                        d.aredistn[(size_t)(nr - 1) * d.padC + cell] = farea * ardg2n;
                        d.vredistn[(size_t)(nr - 1) * d.padC + cell] = fvol * virdgn;
                    }
                }
                s.aicen[nr] = s.aicen[nr] + farea * ardg2n;
                s.vicen[nr] = s.vicen[nr] + fvol * virdgn;
                s.vsnon[nr] = s.vsnon[nr] + fvol * vsrdgn * RDG_FSNOW;
                frac[((n - 1) * NC + nr - 1) * RIDGE_BLOCK] = farea;
                frac[(NC * NC + (n - 1) * NC + nr - 1) * RIDGE_BLOCK] = fvol;
            }
        }
    }
    // The tracers.  Where nothing ridges the products are formed and divided again all the same (:1151, :1530): a
    // category below puny then gets its tracers reset (colpkg_compute_tracers).  The next tracer's values are requested
    // before this one's are stored.
    double tvNext[NC + 1];
    if (ntrcr > 0) FORCAT(n) tvNext[n] = trc(d, 0, n, cell);
    for (int it = 0; it < ntrcr; it++) {
        const int dep = d.tr[it].depend;
        const double b1 = dep == 0 ? 1.0 : 0.0, b2 = dep == 1 ? 1.0 : 0.0, b3 = dep == 2 ? 1.0 : 0.0;
        double tv[NC + 1], at[NC + 1];
        FORCAT(n) tv[n] = tvNext[n];
        if (it + 1 < ntrcr) FORCAT(n) tvNext[n] = trc(d, it + 1, n, cell);
        FORCAT(n) at[n] = tv[n] * (b1 * aicen_init[n] + b2 * vicen_init[n] + b3 * vsnon_init[n]);
        // The reference forms  work = 0 + f1 * tv + f2 * tv + f3 * tv  with f1 = trcr_base(it,1) * ardg2n * farea etc.
        // (:1508-1517).  With a unit trcr_base two of the three products are zeros and the third factor is the exact
        // ardg2n * farea (or virdgn * fvol, or vsrdgn * fvol * fsnowrdg), so 0 + f * tv has the same bits, the sign of a
        // zero included (0 + x turns -0 into +0 as the long sum does).
        if (any) FORCAT(n)
        if (moves[n]) {
            const double out = dep == 0 ? s.ardg1n[n] : dep == 1 ? s.virdgn[n] : vsrdgnC[n];
            const double in = dep == 0 ? s.ardg2n[n] : dep == 1 ? s.virdgn[n] : vsrdgnC[n];
            at[n] = at[n] - (0.0 + out * tv[n]);
            FORCAT(nr)
            {
                const double fr = frac[((dep == 0 ? 0 : NC * NC) + (n - 1) * NC + nr - 1) * RIDGE_BLOCK];
                double f = in * fr;
                if (dep == 2) f = f * RDG_FSNOW;
                at[nr] = at[nr] + (0.0 + f * tv[n]);
            }
        }
        // colpkg_compute_tracers, ice_colpkg_tracers.F90:216-252 (no strata; the loop over the three divisors keeps the
        // last positive one, which with a unit trcr_base is the tracer's own base quantity)
        FORCAT(n)
        {
            const double an = s.aicen[n], vn = s.vicen[n], sn = s.vsnon[n];  // (read before the branches on purpose)
            double v;
            if (dep == 0) {
                if (an > PUNY)
                    v = at[n] / an;
                else
                    v = it == d.ntTsfc ? Tf : 0.0;
            } else {
                const double div = dep == 1 ? vn : sn;
                v = div > 0.0 ? at[n] / div : 0.0;
            }
            trc(d, it, n, cell) = v;
        }
    }
    return 0;
}

// ---- the ITD clean-up, src/column/ice_itd.F90 (constants: constants/cice/ice_constants_colpkg.F90:24-77,
// ice_therm_shared.F90:27)
#define ITD_RHOI 917.0
#define ITD_CP_ICE 2106.0
#define ITD_LFRESH (2.835e6 - 2.501e6)
#define ITD_REF_SALINITY 4.0
#define ITD_HS_MIN 1.0e-4
#define ITD_TMIN (-100.0)

// colpkg_compute_tracers for category n (ice_colpkg_tracers.F90:216-252, no strata)
template <int NC>
__device__ __forceinline__ void computeTracers(const RidgeDev &d, int n, int cell, const RidgeCell<NC> &s, double Tf)
{
    for (int it = 0; it < d.ntrcr; it++) {
        const int dep = d.tr[it].depend;
        const double a = atr(d, it, n, cell);
        const double an = s.aicen[n], vn = s.vicen[n], sn = s.vsnon[n];  // (read before the branches on purpose)
        double v;
        if (dep == 0) {
            if (an > PUNY)
                v = a / an;
            else
                v = it == d.ntTsfc ? Tf : 0.0;
        } else {
            const double div = dep == 1 ? vn : sn;
            v = div > 0.0 ? a / div : 0.0;
        }
        trc(d, it, n, cell) = v;
    }
}

// shift_ice :347-665 with one active boundary n (donor category nd), as rebin calls it.  Returns true for one of
// its four stops; nothing has moved then.
template <int NC>
__device__ __forceinline__ bool shiftIce(const RidgeDev &d, int ncat, int cell, RidgeCell<NC> &s, double (&hicen)[NC + 1], int n, int nd,
                         double daice, double dvice, double Tf)
{
    FORCAT(m)
        for (int it = 0; it < d.ntrcr; it++) {
            const int dep = d.tr[it].depend;
            const double b1 = dep == 0, b2 = dep == 1, b3 = dep == 2;
            atr(d, it, m, cell) = trc(d, it, m, cell) * (b1 * s.aicen[m] + b2 * s.vicen[m] + b3 * s.vsnon[m]);
        }
    bool daiceNeg = false, dviceNeg = false, daiceGreater = false, dviceGreater = false;
    if (daice < 0.0) {
        if (daice > -PUNY * s.aicen[nd]) {
            daice = 0.0;
            dvice = 0.0;
        } else
            daiceNeg = true;
    }
    if (dvice < 0.0) {
        if (dvice > -PUNY * s.vicen[nd]) {
            daice = 0.0;
            dvice = 0.0;
        } else
            dviceNeg = true;
    }
    if (daice > s.aicen[nd] * (1.0 - PUNY)) {
        if (daice < s.aicen[nd] * (1.0 + PUNY)) {
            daice = s.aicen[nd];
            dvice = s.vicen[nd];
        } else
            daiceGreater = true;
    }
    if (dvice > s.vicen[nd] * (1.0 - PUNY)) {
        if (dvice < s.vicen[nd] * (1.0 + PUNY)) {
            daice = s.aicen[nd];
            dvice = s.vicen[nd];
        } else
            dviceGreater = true;
    }
    if (daiceNeg && daice <= -PUNY * s.aicen[nd]) return true;
    if (dviceNeg && dvice <= -PUNY * s.vicen[nd]) return true;
    if (daiceGreater && daice >= s.aicen[nd] * (1.0 + PUNY)) return true;
    if (dviceGreater && dvice >= s.vicen[nd] * (1.0 + PUNY)) return true;
    if (daice > 0.0) {
        const double worka = daice / s.aicen[nd];
        const int nr = nd == n ? nd + 1 : n;
        s.aicen[nd] = s.aicen[nd] - daice;
        s.aicen[nr] = s.aicen[nr] + daice;
        s.vicen[nd] = s.vicen[nd] - dvice;
        s.vicen[nr] = s.vicen[nr] + dvice;
        const double dvsnow = s.vsnon[nd] * worka;
        s.vsnon[nd] = s.vsnon[nd] - dvsnow;
        s.vsnon[nr] = s.vsnon[nr] + dvsnow;
        const double workb = dvsnow;
        for (int it = 0; it < d.ntrcr; it++) {
            const int dep = d.tr[it].depend;
            const double b1 = dep == 0, b2 = dep == 1, b3 = dep == 2;
            const double datrcr = trc(d, it, nd, cell) * (b1 * daice + b2 * dvice + b3 * workb);
            atr(d, it, nd, cell) = atr(d, it, nd, cell) - datrcr;
            atr(d, it, nr, cell) = atr(d, it, nr, cell) + datrcr;
        }
    }
    FORCAT(m) {
        hicen[m] = s.aicen[m] > PUNY ? s.vicen[m] / s.aicen[m] : 0.0;
        computeTracers<NC>(d, m, cell, s, Tf);
    }
    return false;
}

// zap_snow :1340-1422 for category n
template <int NC>
__device__ __forceinline__ void zapSnow(const RidgeDev &d, int n, int cell, RidgeCell<NC> &s, double dt, double &dfresh, double &dfhocn)
{
    if (d.ntQsno >= 0)
        for (int k = 1; k <= d.nslyr; k++) {
            const double xtmp = trc(d, d.ntQsno + k - 1, n, cell) / dt * s.vsnon[n] / (double)d.nslyr;
            dfhocn = dfhocn + xtmp;
            trc(d, d.ntQsno + k - 1, n, cell) = 0.0;
        }
    const double xtmp = (RDG_RHOS * s.vsnon[n]) / dt;
    dfresh = dfresh + xtmp;
    s.vsnon[n] = 0.0;
}

// cleanup_itd :754-1020 for one cell: aggregate_area :53-81, rebin :89-280, zap_small_areas :1029-1336,
// zap_snow_temperature :1426-1559.  Returns true for a stop (colpkg_step_ridge reports them all under one label).
template <int NC>
__device__ __forceinline__ bool cleanupItd(const RidgeDev &d, int ncat, int cell, RidgeCell<NC> &s, double &aice0, double Tf)
{
    const double dt = d.dt * d.ndtd;
    double dfresh = 0.0, dfsalt = 0.0, dfhocn = 0.0;
    double aice = 0.0;
    FORCAT(n) aice = aice + s.aicen[n];
    aice0 = fmax(1.0 - aice, 0.0);
    d.aice[cell] = aice;
    if (aice > 1.0 + PUNY || aice < -PUNY) return true;  // 'aggregate ice area out of bounds'
    if (aice > PUNY) {  // rebin
        double hicen[NC + 1];
        FORCAT(n) hicen[n] = s.aicen[n] > PUNY ? s.vicen[n] / s.aicen[n] : 0.0;
        if (s.aicen[1] > PUNY && hicen[1] <= d.hinMax[0] && d.hinMax[0] > 0.0) {
            s.aicen[1] = s.vicen[1] / d.hinMax[0];
            hicen[1] = d.hinMax[0];
        }
        // a stop inside shift_ice is not looked at by cleanup_itd: zap_small_areas clears the flag (:1102)
        _Pragma("unroll") for (int n = 1; n <= NC - 1; n++) if (n <= ncat - 1)
            if (s.aicen[n] > PUNY && hicen[n] > d.hinMax[n]) shiftIce<NC>(d, ncat, cell, s, hicen, n, n, s.aicen[n], s.vicen[n], Tf);
        _Pragma("unroll") for (int n = NC - 1; n >= 1; n--) if (n <= ncat - 1)
            if (s.aicen[n + 1] > PUNY && hicen[n + 1] <= d.hinMax[n])
                shiftIce<NC>(d, ncat, cell, s, hicen, n, n + 1, s.aicen[n + 1], s.vicen[n + 1], Tf);
    }
    // zap_small_areas
    FORCAT(n) {
        if (s.aicen[n] < -PUNY)
            return true;  // 'Zap ice: negative ice area'
        else if (fabs(s.aicen[n]) != 0.0 && fabs(s.aicen[n]) <= PUNY) {
            if (d.ntQice >= 0)
                for (int k = 1; k <= d.nilyr; k++) {
                    const double xtmp = trc(d, d.ntQice + k - 1, n, cell) / dt * s.vicen[n] / (double)d.nilyr;
                    dfhocn = dfhocn + xtmp;
                    trc(d, d.ntQice + k - 1, n, cell) = 0.0;
                }
            double xtmp = (ITD_RHOI * s.vicen[n]) / dt;
            dfresh = dfresh + xtmp;
            xtmp = ITD_RHOI * s.vicen[n] * ITD_REF_SALINITY * 0.001 / dt;
            dfsalt = dfsalt + xtmp;
            aice0 = aice0 + s.aicen[n];
            s.aicen[n] = 0.0;
            s.vicen[n] = 0.0;
            if (d.ntTsfc >= 0) trc(d, d.ntTsfc, n, cell) = Tf;
            zapSnow<NC>(d, n, cell, s, dt, dfresh, dfhocn);
            for (int it = 1; it < d.ntrcr; it++) trc(d, it, n, cell) = 0.0;  // tracers 2..ntrcr (:1203-1211), no brine
            d.firstIce[(size_t)(n - 1) * d.padC + cell] = 1;
        }
    }
    if (aice > 1.0 + PUNY)
        return true;  // 'Zap ice: excess ice area'
    else if (aice > 1.0 && aice < 1.0 + PUNY) {
        FORCAT(n) {
            if (d.ntQice >= 0)
                for (int k = 1; k <= d.nilyr; k++) {
                    const double xtmp = trc(d, d.ntQice + k - 1, n, cell) * s.vicen[n] / (double)d.nilyr * (aice - 1.0) / aice / dt;
                    dfhocn = dfhocn + xtmp;
                }
            if (d.ntQsno >= 0)
                for (int k = 1; k <= d.nslyr; k++) {
                    const double xtmp = trc(d, d.ntQsno + k - 1, n, cell) * s.vsnon[n] / (double)d.nslyr * (aice - 1.0) / aice / dt;
                    dfhocn = dfhocn + xtmp;
                }
            double xtmp = (ITD_RHOI * s.vicen[n] + RDG_RHOS * s.vsnon[n]) * (aice - 1.0) / aice / dt;
            dfresh = dfresh + xtmp;
            xtmp = ITD_RHOI * s.vicen[n] * ITD_REF_SALINITY * 0.001 * (aice - 1.0) / aice / dt;
            dfsalt = dfsalt + xtmp;
            s.aicen[n] = s.aicen[n] * (1.0 / aice);
            s.vicen[n] = s.vicen[n] * (1.0 / aice);
            s.vsnon[n] = s.vsnon[n] * (1.0 / aice);
        }
        aice = 1.0;
        aice0 = 0.0;
    }
    d.aice[cell] = aice;
    // zap_snow_temperature
    if (d.ntQsno >= 0) {
        const double rnslyr = (double)d.nslyr;
        FORCAT(n) {
            bool zap = false;
            if (s.aicen[n] > PUNY) {
                const double hsn = s.vsnon[n] / s.aicen[n];
                for (int k = 1; k <= d.nslyr; k++) {
                    double zqsn, Tmax;
                    if (hsn > ITD_HS_MIN) {
                        zqsn = trc(d, d.ntQsno + k - 1, n, cell);
                        Tmax = -zqsn * PUNY * rnslyr / (RDG_RHOS * ITD_CP_ICE * s.vsnon[n]);
                    } else {
                        zqsn = -RDG_RHOS * ITD_LFRESH;
                        Tmax = PUNY;
                    }
                    const double zTsn = (ITD_LFRESH + zqsn / RDG_RHOS) / ITD_CP_ICE;
                    if (zTsn < ITD_TMIN || zTsn > Tmax) zap = true;
                }
            }
            if (zap) zapSnow<NC>(d, n, cell, s, dt, dfresh, dfhocn);
        }
    }
    d.fresh[cell] = d.fresh[cell] + dfresh;
    d.fsalt[cell] = d.fsalt[cell] + dfsalt;
    d.fhocn[cell] = d.fhocn[cell] + dfhocn;
    return false;
}

template <int NC>
__global__ void __launch_bounds__(RIDGE_BLOCK, 2) k_ridge_ice(RidgeDev d)
{
    __shared__ double fracStore[2 * NC * NC * RIDGE_BLOCK];
    RidgeLdsDouble *frac = (RidgeLdsDouble *)fracStore + threadIdx.x;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= d.nCellsSolve) return;
    const int ncat = d.ncat, P = d.padC;
    RidgeCell<NC> s;
    FORCAT(n) {
        s.aicen[n] = d.aicen[(size_t)(n - 1) * P + cell];
        s.vicen[n] = d.vicen[(size_t)(n - 1) * P + cell];
        s.vsnon[n] = d.vsnon[(size_t)(n - 1) * P + cell];
        s.ardg1n[n] = s.ardg2n[n] = s.virdgn[n] = s.mraftn[n] = 0.0;
    }
    double aice0 = d.aice0[cell];
    const double Tf = d.Tf[cell], dt = d.dt;
    double msnow_mlt = 0.0, esnow_mlt = 0.0, ardg1 = 0.0, ardg2 = 0.0, virdg = 0.0, aopen = 0.0;
    // asum_ridging :609-632, ridge_prep :640-722
    double asum = aice0;
    FORCAT(n) asum = asum + s.aicen[n];
    double closing_net = RDG_CS * d.rdgShear[cell] + d.rdgConv[cell];
    double divu_adv = (1.0 - asum) / dt;
    if (divu_adv < 0.0) closing_net = fmax(closing_net, -divu_adv);
    double opning = closing_net + divu_adv;
    const int nitermax = 20;
    int stop = 0, niter;
    for (niter = 1; niter <= nitermax; niter++) {
        double aksum;
        ridgeItd<NC>(d, ncat, aice0, s, aksum);
        stop = ridgeShift<NC>(d, ncat, cell, s, aice0, aksum, closing_net, opning, ardg1, ardg2, virdg, aopen, msnow_mlt,
                              esnow_mlt, Tf, frac);
        if (stop) break;
        asum = aice0;
        FORCAT(n) asum = asum + s.aicen[n];
        bool iterate;
        if (fabs(asum - 1.0) < PUNY) {
            iterate = false;
            closing_net = 0.0;
            opning = 0.0;
        } else {
            iterate = true;
            divu_adv = (1.0 - asum) / dt;
            closing_net = fmax(0.0, -divu_adv);
            opning = fmax(0.0, divu_adv);
        }
        if (!iterate) break;
        if (niter == nitermax) {
            stop = 3;
            break;
        }
    }
    atomicMax(&d.flag[1], niter > nitermax ? nitermax : niter);
    FORCAT(n) {
        const size_t k = (size_t)(n - 1) * P + cell;
        d.aicen[k] = s.aicen[n];
        d.vicen[k] = s.vicen[n];
        d.vsnon[k] = s.vsnon[n];
        d.aparticn[k] = s.apartic[n];
        d.krdgn[k] = s.krdg[n];
    }
    d.aice0[cell] = aice0;
    if (!stop && fabs(asum - 1.0) > PUNY) stop = 5;  // :576 (after the diagnostics in the reference; they are set below)
    if (stop && stop != 5) {
        atomicMax(&d.flag[0], stop);
        return;
    }
    double dti = 1.0 / dt;  // :509-548
    d.dardg1dt[cell] = ardg1 * dti;
    d.dardg2dt[cell] = ardg2 * dti;
    d.dvirdgdt[cell] = virdg * dti;
    d.opening[cell] = aopen * dti;
    FORCAT(n) {
        const size_t k = (size_t)(n - 1) * P + cell;
        d.dardg1ndt[k] = s.ardg1n[n] * dti;
        d.dardg2ndt[k] = s.ardg2n[n] * dti;
        d.dvirdgndt[k] = s.virdgn[n] * dti;
        d.araftn[k] = s.mraftn[n] * s.ardg2n[n];
        d.vraftn[k] = s.mraftn[n] * s.virdgn[n];
    }
    dti = 1.0 / (d.ndtd * dt);  // :555-562
    d.fresh[cell] = d.fresh[cell] + msnow_mlt * dti;
    d.fhocn[cell] = d.fhocn[cell] + esnow_mlt * dti;
    if (stop) {
        atomicMax(&d.flag[0], stop);
        return;
    }
    if (d.cleanup) {  // colpkg_step_ridge :3327-3354
        const bool bad = cleanupItd<NC>(d, ncat, cell, s, aice0, Tf);
        FORCAT(n) {
            const size_t k = (size_t)(n - 1) * P + cell;
            d.aicen[k] = s.aicen[n];
            d.vicen[k] = s.vicen[n];
            d.vsnon[k] = s.vsnon[n];
        }
        d.aice0[cell] = aice0;
        if (bad) atomicMax(&d.flag[0], 6);
    }
}

__global__ void k_ridge_zero_flags(int *f)
{
    if (threadIdx.x < 2) f[threadIdx.x] = 0;
}

static int fieldPtr(sih_ctx *c, const char *name, SihLoc loc, int lead, double **out)
{
    int rc = sih_add_field(c, name, loc, lead, false);  // an existing field of that name is kept
    if (rc) return rc;
    SihField *f = sih_find_field(c, name);
    if (!f || f->lead != lead || f->isInt) {
        sih_set_error(std::string("sih_column_ridge: field ") + name + " exists with another shape");
        return SIH_ERR_ARG;
    }
    *out = (double *)f->d;
    return SIH_OK;
}

// the cell fields of the ridging beyond the tracers, created on first use
static int ridgeFields(sih_ctx *c, int nCat, RidgeDev &d)
{
    int rc;
    double *p = nullptr;
    if ((rc = fieldPtr(c, "openWaterArea", LOC_CELL, 1, &d.aice0))) return rc;
    if ((rc = fieldPtr(c, "ridgeConvergence", LOC_CELL, 1, &p))) return rc;
    d.rdgConv = p;
    if ((rc = fieldPtr(c, "ridgeShear", LOC_CELL, 1, &p))) return rc;
    d.rdgShear = p;
    if ((rc = fieldPtr(c, "seaFreezingTemperature", LOC_CELL, 1, &p))) return rc;
    d.Tf = p;
    if ((rc = fieldPtr(c, "areaLossRidge", LOC_CELL, 1, &d.dardg1dt))) return rc;
    if ((rc = fieldPtr(c, "areaGainRidge", LOC_CELL, 1, &d.dardg2dt))) return rc;
    if ((rc = fieldPtr(c, "iceVolumeRidged", LOC_CELL, 1, &d.dvirdgdt))) return rc;
    if ((rc = fieldPtr(c, "openingRateRidge", LOC_CELL, 1, &d.opening))) return rc;
    if ((rc = fieldPtr(c, "oceanFreshWaterFlux", LOC_CELL, 1, &d.fresh))) return rc;
    if ((rc = fieldPtr(c, "oceanHeatFlux", LOC_CELL, 1, &d.fhocn))) return rc;
    if ((rc = fieldPtr(c, "ridgeParticipationFunction", LOC_CELL, nCat, &d.aparticn))) return rc;
    if ((rc = fieldPtr(c, "ratioRidgeThicknessToIce", LOC_CELL, nCat, &d.krdgn))) return rc;
    if ((rc = fieldPtr(c, "fractionNewRidgeArea", LOC_CELL, nCat, &d.aredistn))) return rc;
    if ((rc = fieldPtr(c, "fractionNewRidgeVolume", LOC_CELL, nCat, &d.vredistn))) return rc;
    if ((rc = fieldPtr(c, "areaLossRidgeCategory", LOC_CELL, nCat, &d.dardg1ndt))) return rc;
    if ((rc = fieldPtr(c, "areaGainRidgeCategory", LOC_CELL, nCat, &d.dardg2ndt))) return rc;
    if ((rc = fieldPtr(c, "iceVolumeRidgedCategory", LOC_CELL, nCat, &d.dvirdgndt))) return rc;
    if ((rc = fieldPtr(c, "raftingIceArea", LOC_CELL, nCat, &d.araftn))) return rc;
    if ((rc = fieldPtr(c, "raftingIceVolume", LOC_CELL, nCat, &d.vraftn))) return rc;
    // what the ITD clean-up adds: the aggregate ice area, the salt flux, the categories it emptied
    if ((rc = fieldPtr(c, "iceAreaCell", LOC_CELL, 1, &d.aice))) return rc;
    if ((rc = fieldPtr(c, "oceanSaltFlux", LOC_CELL, 1, &d.fsalt))) return rc;
    if ((rc = sih_add_field(c, "newlyFormedIce", LOC_CELL, nCat, true))) return rc;
    {
        SihField *f = sih_find_field(c, "newlyFormedIce");
        if (!f || f->lead != nCat || !f->isInt) {
            sih_set_error("sih_column_ridge: field newlyFormedIce exists with another shape");
            return SIH_ERR_ARG;
        }
        d.firstIce = (int *)f->d;
    }
    return SIH_OK;
}

extern "C" int sih_column_ridge_init(sih_ctx *c, int nCategories)
{
    if (!c || nCategories < 1 || nCategories > RIDGE_MAXCAT) {
        sih_set_error("sih_column_ridge_init: bad argument");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    RidgeDev d{};
    int rc = ridgeFields(c, nCategories, d);
    // what sih_column_update_state and sih_column_reinitialize_diagnostics work on besides
    static const char *more[] = {"iceVolumeCell", "snowVolumeCell", "iceAreaTendencyTransport", "iceVolumeTendencyTransport",
                                 "iceAgeTendencyTransport"};
    double *p = nullptr;
    for (size_t i = 0; !rc && i < sizeof more / sizeof more[0]; i++) rc = fieldPtr(c, more[i], LOC_CELL, 1, &p);
    return rc;
}

extern "C" int sih_column_ridge(sih_ctx *c, const sih_ridge_config *cfg)
{
    if (!c || !cfg || !cfg->categoryThicknessLimits) {
        sih_set_error("sih_column_ridge: null argument");
        return SIH_ERR_ARG;
    }
    std::vector<SihTracerInfo> tr;
    int nCat = 0;
    if (!sih_ir_tracer_table(c, tr, &nCat)) {
        sih_set_error("sih_column_ridge: the tracer list is not set (sih_ir_init, sih_ir_set_tracers)");
        return SIH_ERR_STATE;
    }
    if (cfg->nCategories != nCat || nCat < 1 || nCat > RIDGE_MAXCAT) {
        sih_set_error("sih_column_ridge: nCategories must be that of the tracers, at most 10");
        return SIH_ERR_ARG;
    }
    if (!(cfg->dt > 0.0) || cfg->dynamicsSubcycleNumber < 1 || cfg->participation < 0 || cfg->participation > 1 ||
        cfg->redistribution < 0 || cfg->redistribution > 1) {
        sih_set_error("sih_column_ridge: bad time step or option");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    RidgeDev d{};
    int iVol = -1, iSnow = -1;
    for (size_t i = 0; i < tr.size(); i++) {
        if (tr[i].name == "iceVolumeCategory") iVol = (int)i;
        if (tr[i].name == "snowVolumeCategory") iSnow = (int)i;
    }
    if (iVol < 0 || iSnow < 0 || tr[0].L != 1 || tr[iVol].L != 1 || tr[iSnow].L != 1 || tr[iVol].parent != 0 ||
        tr[iSnow].parent != 0) {
        sih_set_error("sih_column_ridge: the tracer list must hold iceVolumeCategory and snowVolumeCategory on the ice area");
        return SIH_ERR_ARG;
    }
    std::vector<RidgeTracer> hostTr;
    d.ntTsfc = -1;
    d.ntQsno = -1;
    d.nslyr = 0;
    d.ntQice = -1;
    d.nilyr = 0;
    d.cleanup = cfg->itdCleanup != 0;
    int nt = 0;
    for (size_t i = 1; i < tr.size(); i++) {
        if ((int)i == iVol || (int)i == iSnow) continue;
        const int dep = tr[i].parent == 0 ? 0 : tr[i].parent == iVol ? 1 : tr[i].parent == iSnow ? 2 : -1;
        if (dep < 0) {
            sih_set_error("sih_column_ridge: tracer " + tr[i].name +
                          " is carried on another tracer (strata: ponds, brine, level ice, aerosols): not supported");
            return SIH_ERR_UNSUPPORTED;
        }
        if (nt + tr[i].L > RIDGE_MAXTR) {
            sih_set_error("sih_column_ridge: too many tracer layers");
            return SIH_ERR_UNSUPPORTED;
        }
        if (tr[i].name == "surfaceTemperature") d.ntTsfc = nt;
        if (tr[i].name == "iceEnthalpy") {
            d.ntQice = nt;
            d.nilyr = tr[i].L;
        }
        if (tr[i].name == "snowEnthalpy") {
            d.ntQsno = nt;
            d.nslyr = tr[i].L;
        }
        for (int l = 0; l < tr[i].L; l++) hostTr.push_back(RidgeTracer{tr[i].val, tr[i].L, l, dep});
        nt += tr[i].L;
    }
    d.ntrcr = nt;
    d.ncat = nCat;
    d.nCellsSolve = c->nCellsSolve;
    d.padC = c->padC;
    d.dt = cfg->dt;
    d.ndtd = cfg->dynamicsSubcycleNumber;
    d.krdgPartic = cfg->participation;
    d.krdgRedist = cfg->redistribution;
    d.muRdg = cfg->efoldingScale;
    for (int n = 0; n <= nCat; n++) d.hinMax[n] = cfg->categoryThicknessLimits[n];
    d.hinMax[nCat] = 1.0e+8;  // ridge_prep :675 (the reference writes it into the host's array as well)
    d.aicen = tr[0].val;
    d.vicen = tr[iVol].val;
    d.vsnon = tr[iSnow].val;
    int rc;
    if ((rc = ridgeFields(c, nCat, d))) return rc;
    if (!c->ridge) c->ridge = new sih_ridge_state();
    sih_ridge_state *rs = c->ridge;
    if (!rs->d_flag) HIPCHK(hipMalloc(&rs->d_flag, 2 * sizeof(int)));
    if (!rs->d_tr) HIPCHK(hipMalloc(&rs->d_tr, RIDGE_MAXTR * sizeof(RidgeTracer)));
    if (rs->hostTr.size() != hostTr.size() ||
        (nt > 0 && memcmp(rs->hostTr.data(), hostTr.data(), hostTr.size() * sizeof(RidgeTracer)) != 0)) {
        HIPCHK(hipStreamSynchronize(c->stream));  // a launch still reading the old table
        if (nt > 0) HIPCHK(hipMemcpy(rs->d_tr, hostTr.data(), hostTr.size() * sizeof(RidgeTracer), hipMemcpyHostToDevice));
        rs->hostTr = hostTr;
    }
    d.tr = rs->d_tr;
    if (rs->planes != nt * nCat) {
        const std::string name = "_ridgeProducts" + std::to_string(nt * nCat);
        if ((rc = fieldPtr(c, name.c_str(), LOC_CELL, nt * nCat, &rs->atrcrn))) return rc;
        rs->planes = nt * nCat;
    }
    d.atrcrn = rs->atrcrn;
    d.flag = rs->d_flag;
    hipLaunchKernelGGL(k_ridge_zero_flags, dim3(1), dim3(64), 0, c->stream, d.flag);
    const dim3 grid((unsigned)((d.nCellsSolve + RIDGE_BLOCK - 1) / RIDGE_BLOCK)), block(RIDGE_BLOCK);
    if (nCat <= 5)
        hipLaunchKernelGGL(k_ridge_ice<5>, grid, block, 0, c->stream, d);
    else if (nCat <= 7)
        hipLaunchKernelGGL(k_ridge_ice<7>, grid, block, 0, c->stream, d);
    else
        hipLaunchKernelGGL(k_ridge_ice<RIDGE_MAXCAT>, grid, block, 0, c->stream, d);
    HIPCHK(hipGetLastError());
    int flags[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(flags, d.flag, sizeof flags, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    rs->lastIterations = flags[1];
    if (flags[0]) {
        sih_set_error(flags[0] == 1   ? "Ridging error: aice0 < 0"
                      : flags[0] == 2 ? "Ridging error: ardg > aicen"
                      : flags[0] == 3 ? "ridge_ice: Exceeded max number of ridging iterations"
                      : flags[0] == 5 ? "ridge_ice: total area > 1"
                                      : "ice: ITD cleanup error in colpkg_step_ridge");
        return SIH_ERR_COLUMN;
    }
    return SIH_OK;
}

// ---- seaice_column_update_state (src/shared/mpas_seaice_column.F:4279-4377) -------------------------------------
// seaice_column_aggregate (:4391-4494) calls colpkg_aggregate (src/column/ice_colpkg.F90:3366-3480) for every owned
// cell: sums of the category areas and volumes, the open water, and the cell value of every category tracer
// (colpkg_compute_tracers on the sums of tracer * base quantity); then the transport tendencies (:4352-4355).
// One thread per owned cell on the planes the ridging left.
struct AggTracer {
    const double *val;  // plane 0 of the category field
    double *cell;       // plane 0 of the cell field (<name>Cell)
    int L, layer, depend, isTsfc;
};
struct AggDev {
    int nCellsSolve, padC, ncat, ntrcr, tendencies;
    double dt;
    const double *aicen, *vicen, *vsnon, *Tf;
    double *aice, *vice, *vsno, *aice0, *tendA, *tendV;
    AggTracer tr[RIDGE_MAXTR];
};

template <int NC>
__global__ void __launch_bounds__(256) k_column_aggregate(AggDev d)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= d.nCellsSolve) return;
    const int ncat = d.ncat;
    const size_t P = d.padC;
    double a[NC], v[NC], s[NC];
    double aice = 0.0, vice = 0.0, vsno = 0.0;
    _Pragma("unroll") for (int n = 0; n < NC; n++) if (n < ncat) {
        a[n] = d.aicen[n * P + cell];
        v[n] = d.vicen[n * P + cell];
        s[n] = d.vsnon[n * P + cell];
        aice = aice + a[n];
        vice = vice + v[n];
        vsno = vsno + s[n];
    }
    d.aice[cell] = aice;
    d.vice[cell] = vice;
    d.vsno[cell] = vsno;
    d.aice0[cell] = fmax(1.0 - aice, 0.0);
    const double Tf = d.Tf[cell];
    for (int it = 0; it < d.ntrcr; it++) {
        const AggTracer t = d.tr[it];
        const double b1 = t.depend == 0 ? 1.0 : 0.0, b2 = t.depend == 1 ? 1.0 : 0.0, b3 = t.depend == 2 ? 1.0 : 0.0;
        double atrcr = 0.0;
        _Pragma("unroll") for (int n = 0; n < NC; n++) if (n < ncat) {
            const double atrcrn = t.val[((size_t)n * t.L + t.layer) * P + cell] * (b1 * a[n] + b2 * v[n] + b3 * s[n]);
            atrcr = atrcr + atrcrn;
        }
        double out;  // colpkg_compute_tracers, ice_colpkg_tracers.F90:163-254, trcr_base the unit vector of the base
        if (t.depend == 0) {
            if (aice > PUNY)
                out = atrcr / aice;
            else
                out = t.isTsfc ? Tf : 0.0;
        } else {
            const double divisor = t.depend == 1 ? vice : vsno;
            out = divisor > 0.0 ? atrcr / divisor : 0.0;
        }
        t.cell[(size_t)t.layer * P + cell] = out;
    }
    if (d.tendencies) {  // :4352-4355, stateUpdateType "transport"
        d.tendA[cell] = (aice - d.tendA[cell]) / d.dt;
        d.tendV[cell] = (vice - d.tendV[cell]) / d.dt;
    }
}

// seaice_column_reinitialize_diagnostics_dynamics (:13680-13718), the fields of this path: whole arrays
__global__ void k_column_reinit_diagnostics(int n, double *r0, double *r1, double *r2, double *r3, const double *aice,
                                            const double *vice, double *tendA, double *tendV, double *tendAge)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    r0[i] = 0.0;
    r1[i] = 0.0;
    r2[i] = 0.0;
    r3[i] = 0.0;
    tendA[i] = aice[i];
    tendV[i] = vice[i];
    tendAge[i] = 0.0;
}

static int aggregateSetup(sih_ctx *c, const char *who, int nCategories, AggDev &d)
{
    std::vector<SihTracerInfo> tr;
    int nCat = 0;
    if (!sih_ir_tracer_table(c, tr, &nCat)) {
        sih_set_error(std::string(who) + ": the tracer list is not set (sih_ir_init, sih_ir_set_tracers)");
        return SIH_ERR_STATE;
    }
    if (nCategories != nCat || nCat < 1 || nCat > RIDGE_MAXCAT) {
        sih_set_error(std::string(who) + ": nCategories must be that of the tracers, at most 10");
        return SIH_ERR_ARG;
    }
    int iVol = -1, iSnow = -1;
    for (size_t i = 0; i < tr.size(); i++) {
        if (tr[i].name == "iceVolumeCategory") iVol = (int)i;
        if (tr[i].name == "snowVolumeCategory") iSnow = (int)i;
    }
    if (iVol < 0 || iSnow < 0 || tr[0].L != 1 || tr[iVol].L != 1 || tr[iSnow].L != 1 || tr[iVol].parent != 0 ||
        tr[iSnow].parent != 0) {
        sih_set_error(std::string(who) + ": the tracer list must hold iceVolumeCategory and snowVolumeCategory on the ice area");
        return SIH_ERR_ARG;
    }
    int rc, nt = 0;
    for (size_t i = 1; i < tr.size(); i++) {
        if ((int)i == iVol || (int)i == iSnow) continue;
        const int dep = tr[i].parent == 0 ? 0 : tr[i].parent == iVol ? 1 : tr[i].parent == iSnow ? 2 : -1;
        if (dep < 0) {
            sih_set_error(std::string(who) + ": tracer " + tr[i].name +
                          " is carried on another tracer (strata: ponds, brine, level ice, aerosols): not supported");
            return SIH_ERR_UNSUPPORTED;
        }
        if (nt + tr[i].L > RIDGE_MAXTR) {
            sih_set_error(std::string(who) + ": too many tracer layers");
            return SIH_ERR_UNSUPPORTED;
        }
        double *cell = nullptr;  // the cell field of the `tracers_aggregate` pool: <tracer>Cell, the tracer's layers deep
        if ((rc = fieldPtr(c, (tr[i].name + "Cell").c_str(), LOC_CELL, tr[i].L, &cell))) return rc;
        for (int l = 0; l < tr[i].L; l++)
            d.tr[nt++] = AggTracer{tr[i].val, cell, tr[i].L, l, dep, tr[i].name == "surfaceTemperature" ? 1 : 0};
    }
    d.ntrcr = nt;
    d.ncat = nCat;
    d.nCellsSolve = c->nCellsSolve;
    d.padC = c->padC;
    d.aicen = tr[0].val;
    d.vicen = tr[iVol].val;
    d.vsnon = tr[iSnow].val;
    double *p = nullptr;
    if ((rc = fieldPtr(c, "seaFreezingTemperature", LOC_CELL, 1, &p))) return rc;
    d.Tf = p;
    if ((rc = fieldPtr(c, "iceAreaCell", LOC_CELL, 1, &d.aice))) return rc;
    if ((rc = fieldPtr(c, "iceVolumeCell", LOC_CELL, 1, &d.vice))) return rc;
    if ((rc = fieldPtr(c, "snowVolumeCell", LOC_CELL, 1, &d.vsno))) return rc;
    if ((rc = fieldPtr(c, "openWaterArea", LOC_CELL, 1, &d.aice0))) return rc;
    if ((rc = fieldPtr(c, "iceAreaTendencyTransport", LOC_CELL, 1, &d.tendA))) return rc;
    if ((rc = fieldPtr(c, "iceVolumeTendencyTransport", LOC_CELL, 1, &d.tendV))) return rc;
    return SIH_OK;
}

extern "C" int sih_column_update_state(sih_ctx *c, int nCategories, double dt, int tendencies)
{
    if (!c || (tendencies && !(dt > 0.0))) {
        sih_set_error("sih_column_update_state: bad argument");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    AggDev d{};
    int rc;
    if ((rc = aggregateSetup(c, "sih_column_update_state", nCategories, d))) return rc;
    d.dt = dt;
    d.tendencies = tendencies != 0;
    const dim3 grid((unsigned)((d.nCellsSolve + 255) / 256)), block(256);
    if (d.ncat <= 5)
        hipLaunchKernelGGL(k_column_aggregate<5>, grid, block, 0, c->stream, d);
    else
        hipLaunchKernelGGL(k_column_aggregate<RIDGE_MAXCAT>, grid, block, 0, c->stream, d);
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

extern "C" int sih_column_reinitialize_diagnostics(sih_ctx *c, int nCategories)
{
    if (!c) {
        sih_set_error("sih_column_reinitialize_diagnostics: null argument");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    AggDev d{};
    RidgeDev r{};
    int rc;
    if ((rc = aggregateSetup(c, "sih_column_reinitialize_diagnostics", nCategories, d))) return rc;
    if ((rc = ridgeFields(c, nCategories, r))) return rc;
    double *age = nullptr;
    if ((rc = fieldPtr(c, "iceAgeTendencyTransport", LOC_CELL, 1, &age))) return rc;
    const int n = c->nCells + 1;
    hipLaunchKernelGGL(k_column_reinit_diagnostics, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, r.dardg1dt,
                       r.dardg2dt, r.dvirdgdt, r.opening, d.aice, d.vice, d.tendA, d.tendV, age);
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

double sih_ridge_info(sih_ctx *c) { return c->ridge ? (double)c->ridge->lastIterations : 0.0; }
