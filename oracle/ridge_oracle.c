/* CPU oracle for the mechanical redistribution (ridging) of the column package: a plain-C restatement of
 * ridge_ice and the routines it calls in /root/reference/src/column/ice_mechred.F90 (asum_ridging :609, ridge_prep
 * :640, ridge_itd :741, ridge_shift :1030) and of colpkg_compute_tracers
 * (/root/reference/src/column/ice_colpkg_tracers.F90:163-254), as column_ridging drives them for every owned cell
 * (/root/reference/src/shared/mpas_seaice_column.F:3468-3530 -> colpkg_step_ridge,
 * /root/reference/src/column/ice_colpkg.F90:3296-3325).  Loops and expressions keep the reference's order.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under mpas_seaice_amd/ may call this.  Built with -ffp-contract=off.
 * PARITY UNPINNED: the reference holds no vectors for this path and cannot be built here (it needs the column
 * package's tracer bookkeeping and the MPAS framework); the restatement is pinned by conservation of ice and snow
 * volume and of the volume-weighted tracers, by asum = 1 after ridging, and by closed forms for one category.
 *
 * Scope: the tracers of the standard physics set without strata -- surface temperature (on ice area), ice enthalpy
 * and salinity per ice layer (on ice volume), snow enthalpy per snow layer (on snow volume); n_trcr_strata = 0 for
 * all of them, so the strata branches (:1206-1211, :1389-1394, :1508-1517; ponds, brine, aerosols, level ice) are
 * not restated.  The ITD clean-up that colpkg_step_ridge runs after ridge_ice (cleanup_itd, ice_itd.F90:754) follows
 * in the second half of this file.  The module's l_conservation_check is .false. (:64): that block (:299-334,
 * :430-503) is not run. */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define MAXCAT 16
#define MAXTR 64

static const double puny = 1.0e-11, c0 = 0.0, c1 = 1.0, c2 = 2.0, p5 = 0.5;
/* ice_mechred.F90:51-61 */
static const double Cs = 0.25, fsnowrdg = 0.5, Gstar = 0.15, astar = 0.05, maxraft = 1.0, Hstar = 25.0;
static const double rhos = 330.0; /* ice_constants_colpkg.F90:24 */

/* ridge_itd :741-1012 (1-based category arrays of size ncat+1; apartic and Gsum from 0 / -1) */
static void ridge_itd(int ncat, double aice0, const double *aicen, const double *vicen, int krdg_partic, int krdg_redist,
                      double mu_rdg, double *aksum, double *apartic, double *hrmin, double *hrmax, double *hrexp,
                      double *krdg, double *mraft)
{
    const double Gstari = c1 / Gstar, astari = c1 / astar;
    double GsumA[MAXCAT + 2];
    double *Gsum = GsumA + 1;
    Gsum[-1] = c0;
    if (aice0 > puny)
        Gsum[0] = aice0;
    else
        Gsum[0] = Gsum[-1];
    apartic[0] = c0;
    for (int n = 1; n <= ncat; n++) {
        Gsum[n] = c1;
        apartic[n] = c0;
        hrmin[n] = c0;
        hrmax[n] = c0;
        hrexp[n] = c0;
        krdg[n] = c1;
        if (aicen[n] > puny)
            Gsum[n] = Gsum[n - 1] + aicen[n];
        else
            Gsum[n] = Gsum[n - 1];
    }
    double work = c1 / Gsum[ncat];
    for (int n = 0; n <= ncat; n++) Gsum[n] = Gsum[n] * work;
    if (krdg_partic == 0) {
        for (int n = 0; n <= ncat; n++) {
            if (Gsum[n] < Gstar)
                apartic[n] = Gstari * (Gsum[n] - Gsum[n - 1]) * (c2 - Gstari * (Gsum[n - 1] + Gsum[n]));
            else if (Gsum[n - 1] < Gstar)
                apartic[n] = Gstari * (Gstar - Gsum[n - 1]) * (c2 - Gstari * (Gstar + Gsum[n - 1]));
        }
    } else if (krdg_partic == 1) {
        double xtmp = c1 / (c1 - exp(-astari));
        Gsum[-1] = exp(-Gsum[-1] * astari) * xtmp;
        for (int n = 0; n <= ncat; n++) {
            Gsum[n] = exp(-Gsum[n] * astari) * xtmp;
            apartic[n] = Gsum[n - 1] - Gsum[n];
        }
    }
    if (krdg_redist == 0) {
        for (int n = 1; n <= ncat; n++)
            if (aicen[n] > puny) {
                double hi = vicen[n] / aicen[n];
                hrmin[n] = fmin(c2 * hi, hi + maxraft);
                hrmax[n] = c2 * sqrt(Hstar * hi);
                hrmax[n] = fmax(hrmax[n], hrmin[n] + puny);
                double hrmean = p5 * (hrmin[n] + hrmax[n]);
                krdg[n] = hrmean / hi;
            }
    } else {
        for (int n = 1; n <= ncat; n++)
            if (aicen[n] > puny) {
                double hi = vicen[n] / aicen[n];
                hi = fmax(hi, puny);
                hrmin[n] = fmin(c2 * hi, hi + maxraft);
                hrexp[n] = mu_rdg * sqrt(hi);
                krdg[n] = (hrmin[n] + hrexp[n]) / hi;
                /* diagnostic rafting mask :976-980 */
                mraft[n] = fmax(c0, copysign(c1, hi + maxraft - hrmin[n]));
                double xtmp = mraft[n] * ((c2 * hi + hrexp[n]) / hi - krdg[n]);
                mraft[n] = fmax(c0, copysign(c1, puny - fabs(xtmp)));
            }
    }
    *aksum = apartic[0];
    for (int n = 1; n <= ncat; n++) *aksum = *aksum + apartic[n] * (c1 - c1 / krdg[n]);
}

/* colpkg_compute_tracers, ice_colpkg_tracers.F90:163-254, without strata; trcr_base(it,:) is the unit vector of
 * trcr_depend(it).  The loop over the three divisors keeps the last positive one: with a unit trcr_base that is the
 * one of the tracer's own base quantity. */
static void compute_tracers(int ntrcr, const int *trcr_depend, const double *atrcrn, double aicen, double vicen,
                            double vsnon, int nt_Tsfc, double Tf, double *trcrn)
{
    for (int it = 0; it < ntrcr; it++) {
        double divisor[3];
        divisor[0] = (trcr_depend[it] == 0 ? 1.0 : 0.0) * aicen;
        divisor[1] = (trcr_depend[it] == 1 ? 1.0 : 0.0) * vicen;
        divisor[2] = (trcr_depend[it] == 2 ? 1.0 : 0.0) * vsnon;
        if (trcr_depend[it] == 0) {
            if (aicen > puny)
                trcrn[it] = atrcrn[it] / aicen;
            else {
                trcrn[it] = c0;
                if (it == nt_Tsfc) trcrn[it] = Tf;
            }
        } else {
            double work = c0;
            for (int k = 0; k < 3; k++)
                if (divisor[k] > c0) work = atrcrn[it] / divisor[k];
            trcrn[it] = work;
        }
    }
}

/* ridge_shift :1030-1541.  trcrn is (ntrcr, ncat), tracer index fastest, categories 1..ncat at column n - 1.
 * Returns 0, or the number of the reference's stop: 1 'Ridging error: aice0 < 0', 2 'Ridging error: ardg > aicen'. */
static int ridge_shift(int ntrcr, double dt, int ncat, const double *hin_max, double *aicen, double *trcrn, double *vicen,
                       double *vsnon, double *aice0, const int *trcr_depend, int krdg_redist, double aksum,
                       const double *apartic, const double *hrmin, const double *hrmax, const double *hrexp,
                       const double *krdg, double *closing_net, double *opning, double *ardg1, double *ardg2,
                       double *virdg, double *aopen, double *ardg1nn, double *ardg2nn, double *virdgnn, int nslyr,
                       int nt_qsno, int nt_Tsfc, double *msnow_mlt, double *esnow_mlt, double *aredistn, double *vredistn,
                       double Tf)
{
    double aicen_init[MAXCAT + 1], vicen_init[MAXCAT + 1], vsnon_init[MAXCAT + 1];
    double atrcrn[MAXCAT + 1][MAXTR];
    (void)closing_net;
    for (int n = 1; n <= ncat; n++) {
        aicen_init[n] = aicen[n];
        vicen_init[n] = vicen[n];
        vsnon_init[n] = vsnon[n];
        for (int it = 0; it < ntrcr; it++) {
            const double b1 = trcr_depend[it] == 0, b2 = trcr_depend[it] == 1, b3 = trcr_depend[it] == 2;
            atrcrn[n][it] = trcrn[(size_t)(n - 1) * ntrcr + it] * (b1 * aicen[n] + b2 * vicen[n] + b3 * vsnon[n]);
        }
    }
    double closing_gross = *closing_net / aksum;
    if (apartic[0] > c0) {
        double wk1 = apartic[0] * closing_gross * dt;
        if (wk1 > *aice0) {
            double tmpfac = *aice0 / wk1;
            closing_gross = closing_gross * tmpfac;
            *opning = *opning * tmpfac;
        }
    }
    for (int n = 1; n <= ncat; n++)
        if (aicen[n] > puny && apartic[n] > c0) {
            double wk1 = apartic[n] * closing_gross * dt;
            if (wk1 > aicen[n]) {
                double tmpfac = aicen[n] / wk1;
                closing_gross = closing_gross * tmpfac;
                *opning = *opning * tmpfac;
            }
        }
    *aice0 = *aice0 - apartic[0] * closing_gross * dt + *opning * dt;
    if (*aice0 < -puny)
        return 1;
    else if (*aice0 < c0)
        *aice0 = c0;
    *aopen = *opning * dt;

    for (int n = 1; n <= ncat; n++) {
        if (aicen_init[n] > puny && apartic[n] > c0 && closing_gross > c0) {
            double ardg1n = apartic[n] * closing_gross * dt;
            if (ardg1n > aicen_init[n] + puny)
                return 2;
            else
                ardg1n = fmin(aicen_init[n], ardg1n);
            double ardg2n = ardg1n / krdg[n];
            double afrac = ardg1n / aicen_init[n];
            double virdgn = vicen_init[n] * afrac;
            double vsrdgn = vsnon_init[n] * afrac;
            aicen[n] = aicen[n] - ardg1n;
            vicen[n] = vicen[n] - virdgn;
            vsnon[n] = vsnon[n] - vsrdgn;
            *ardg1 = *ardg1 + ardg1n;
            *ardg2 = *ardg2 + ardg2n;
            *virdg = *virdg + virdgn;
            ardg1nn[n] = ardg1n;
            ardg2nn[n] = ardg2n;
            virdgnn[n] = virdgn;
            *msnow_mlt = *msnow_mlt + rhos * vsrdgn * (c1 - fsnowrdg);
            double dhr = hrmax[n] - hrmin[n];
            double dhr2 = hrmax[n] * hrmax[n] - hrmin[n] * hrmin[n];
            for (int k = 1; k <= nslyr; k++) {
                double esrdgn = vsrdgn * trcrn[(size_t)(n - 1) * ntrcr + nt_qsno + k - 1] / (double)nslyr;
                *esnow_mlt = *esnow_mlt + esrdgn * (c1 - fsnowrdg);
            }
            for (int it = 0; it < ntrcr; it++) {
                double trfactor[3];
                trfactor[0] = (trcr_depend[it] == 0 ? 1.0 : 0.0) * ardg1n;
                trfactor[1] = (trcr_depend[it] == 1 ? 1.0 : 0.0) * virdgn;
                trfactor[2] = (trcr_depend[it] == 2 ? 1.0 : 0.0) * vsrdgn;
                double work = c0;
                for (int k = 0; k < 3; k++) work = work + trfactor[k] * trcrn[(size_t)(n - 1) * ntrcr + it];
                atrcrn[n][it] = atrcrn[n][it] - work;
            }
            for (int nr = 1; nr <= ncat; nr++) {
                double farea, fvol, hL, hR;
                if (krdg_redist == 0) {
                    if (hrmin[n] >= hin_max[nr] || hrmax[n] <= hin_max[nr - 1]) {
                        hL = c0;
                        hR = c0;
                    } else {
                        hL = fmax(hrmin[n], hin_max[nr - 1]);
                        hR = fmin(hrmax[n], hin_max[nr]);
                    }
                    farea = (hR - hL) / dhr;
                    fvol = (hR * hR - hL * hL) / dhr2;
                } else {
                    if (nr < ncat) {
                        double hi1 = hrmin[n], hexp = hrexp[n];
                        if (hi1 >= hin_max[nr]) {
                            farea = c0;
                            fvol = c0;
                        } else {
                            hL = fmax(hi1, hin_max[nr - 1]);
                            hR = hin_max[nr];
                            double expL = exp(-(hL - hi1) / hexp);
                            double expR = exp(-(hR - hi1) / hexp);
                            farea = expL - expR;
                            fvol = ((hL + hexp) * expL - (hR + hexp) * expR) / (hi1 + hexp);
                        }
                    } else {
                        double hi1 = hrmin[n], hexp = hrexp[n];
                        hL = fmax(hi1, hin_max[nr - 1]);
                        double expL = exp(-(hL - hi1) / hexp);
                        farea = expL;
                        fvol = (hL + hexp) * expL / (hi1 + hexp);
                    }
                    if (n == 1) { /* diagnostics, thinnest ridging ice only :1462-1469 */
                        aredistn[nr] = farea * ardg2n;
                        vredistn[nr] = fvol * virdgn;
                    }
                }
                aicen[nr] = aicen[nr] + farea * ardg2n;
                vicen[nr] = vicen[nr] + fvol * virdgn;
                vsnon[nr] = vsnon[nr] + fvol * vsrdgn * fsnowrdg;
                for (int it = 0; it < ntrcr; it++) {
                    double trfactor[3];
                    trfactor[0] = (trcr_depend[it] == 0 ? 1.0 : 0.0) * ardg2n * farea;
                    trfactor[1] = (trcr_depend[it] == 1 ? 1.0 : 0.0) * virdgn * fvol;
                    trfactor[2] = (trcr_depend[it] == 2 ? 1.0 : 0.0) * vsrdgn * fvol * fsnowrdg;
                    double work = c0;
                    for (int k = 0; k < 3; k++) work = work + trfactor[k] * trcrn[(size_t)(n - 1) * ntrcr + it];
                    atrcrn[nr][it] = atrcrn[nr][it] + work;
                }
            }
        }
    }
    for (int n = 1; n <= ncat; n++)
        compute_tracers(ntrcr, trcr_depend, atrcrn[n], aicen[n], vicen[n], vsnon[n], nt_Tsfc, Tf,
                        trcrn + (size_t)(n - 1) * ntrcr);
    return 0;
}

/* ridge_ice :80-597 for one cell.  Category arrays aicen, vicen, vsnon and the per-category diagnostics are 0-based
 * of length ncat here; hin_max has ncat + 1 entries (0:ncat) and entry ncat is set to 1e8 as ridge_prep does (:675).
 * Returns 0 or the reference's stop: 1, 2 from ridge_shift, 3 'ridge_ice: Exceeded max number of ridging iterations',
 * 5 'ridge_ice: total area > 1'. */
int ora_ridge_ice(double dt, int ndtd, int ncat, int nslyr, int ntrcr, double *hin_max, double rdg_conv, double rdg_shear,
                  double *aicen0, double *trcrn, double *vicen0, double *vsnon0, double *aice0, const int *trcr_depend,
                  int nt_Tsfc, int nt_qsno, int krdg_partic, int krdg_redist, double mu_rdg, double *dardg1dt,
                  double *dardg2dt, double *dvirdgdt, double *opening, double *fresh, double *fhocn, double *aparticn,
                  double *krdgn, double *aredistn0, double *vredistn0, double *dardg1ndt, double *dardg2ndt,
                  double *dvirdgndt, double *araftn, double *vraftn, double Tf, int *nIterations)
{
    double aicen[MAXCAT + 1], vicen[MAXCAT + 1], vsnon[MAXCAT + 1];
    double apartic[MAXCAT + 1], hrmin[MAXCAT + 1], hrmax[MAXCAT + 1], hrexp[MAXCAT + 1], krdg[MAXCAT + 1];
    double ardg1n[MAXCAT + 1], ardg2n[MAXCAT + 1], virdgn[MAXCAT + 1], mraftn[MAXCAT + 1];
    double aredistn[MAXCAT + 1], vredistn[MAXCAT + 1];
    if (ncat > MAXCAT || ntrcr > MAXTR) return -1;
    for (int n = 1; n <= ncat; n++) {
        aicen[n] = aicen0[n - 1];
        vicen[n] = vicen0[n - 1];
        vsnon[n] = vsnon0[n - 1];
        ardg1n[n] = ardg2n[n] = virdgn[n] = mraftn[n] = c0;
        aredistn[n] = aredistn0[n - 1]; /* intent(inout), optional: left alone where nothing ridges */
        vredistn[n] = vredistn0[n - 1];
    }
    double msnow_mlt = c0, esnow_mlt = c0, ardg1 = c0, ardg2 = c0, virdg = c0, aopen = c0;
    double asum, aksum, closing_net, divu_adv, opning;
    const int nitermax = 20;
    int stop = 0;
    /* asum_ridging :609-632 */
    asum = *aice0;
    for (int n = 1; n <= ncat; n++) asum = asum + aicen[n];
    /* ridge_prep :640-722 */
    hin_max[ncat] = 1.0e+8;
    closing_net = Cs * rdg_shear + rdg_conv;
    divu_adv = (c1 - asum) / dt;
    if (divu_adv < c0) closing_net = fmax(closing_net, -divu_adv);
    opning = closing_net + divu_adv;

    int niter;
    for (niter = 1; niter <= nitermax; niter++) {
        ridge_itd(ncat, *aice0, aicen, vicen, krdg_partic, krdg_redist, mu_rdg, &aksum, apartic, hrmin, hrmax, hrexp, krdg,
                  mraftn);
        for (int n = 1; n <= ncat; n++) { /* the optional diagnostics of ridge_itd :1001-1010 */
            aparticn[n - 1] = apartic[n];
            krdgn[n - 1] = krdg[n];
        }
        stop = ridge_shift(ntrcr, dt, ncat, hin_max, aicen, trcrn, vicen, vsnon, aice0, trcr_depend, krdg_redist, aksum,
                           apartic, hrmin, hrmax, hrexp, krdg, &closing_net, &opning, &ardg1, &ardg2, &virdg, &aopen, ardg1n,
                           ardg2n, virdgn, nslyr, nt_qsno, nt_Tsfc, &msnow_mlt, &esnow_mlt, aredistn, vredistn, Tf);
        if (stop) break;
        asum = *aice0;
        for (int n = 1; n <= ncat; n++) asum = asum + aicen[n];
        int iterate_ridging;
        if (fabs(asum - c1) < puny) {
            iterate_ridging = 0;
            closing_net = c0;
            opning = c0;
        } else {
            iterate_ridging = 1;
            divu_adv = (c1 - asum) / dt;
            closing_net = fmax(c0, -divu_adv);
            opning = fmax(c0, divu_adv);
        }
        if (!iterate_ridging) break;
        if (niter == nitermax) {
            stop = 3;
            break;
        }
    }
    if (nIterations) *nIterations = niter > nitermax ? nitermax : niter;
    for (int n = 1; n <= ncat; n++) {
        aicen0[n - 1] = aicen[n];
        vicen0[n - 1] = vicen[n];
        vsnon0[n - 1] = vsnon[n];
        aredistn0[n - 1] = aredistn[n];
        vredistn0[n - 1] = vredistn[n];
    }
    if (stop) return stop;
    /* diagnostics :509-548 */
    double dti = c1 / dt;
    *dardg1dt = ardg1 * dti;
    *dardg2dt = ardg2 * dti;
    *dvirdgdt = virdg * dti;
    *opening = aopen * dti;
    for (int n = 1; n <= ncat; n++) {
        dardg1ndt[n - 1] = ardg1n[n] * dti;
        dardg2ndt[n - 1] = ardg2n[n] * dti;
        dvirdgndt[n - 1] = virdgn[n] * dti;
        araftn[n - 1] = mraftn[n] * ardg2n[n];
        vraftn[n - 1] = mraftn[n] * virdgn[n];
    }
    /* fluxes :555-562 */
    dti = c1 / (ndtd * dt);
    *fresh = *fresh + msnow_mlt * dti;
    *fhocn = *fhocn + esnow_mlt * dti;
    if (fabs(asum - c1) > puny) return 5;
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * cleanup_itd (ice_itd.F90:754-1020), which colpkg_step_ridge runs after ridge_ice (ice_colpkg.F90:3327-3350), with
 * the routines it calls, for the standard physics tracers (no aerosols, ponds, brine, bio tracers, zsalinity;
 * heat_capacity true; limit_aice true): aggregate_area :53-81, rebin :89-280, shift_ice :347-665, zap_small_areas
 * :1029-1336, zap_snow :1340-1422, zap_snow_temperature :1426-1559.  Category arrays are 1-based (entry 0 unused),
 * trcrn is (ntrcr, ncat) with the tracer index fastest and category n in column n - 1. */
static const double rhoi = 917.0, cp_ice = 2106.0, Lfresh = 2.835e6 - 2.501e6, ice_ref_salinity = 4.0, p001 = 0.001;
static const double hs_min = 1.0e-4, Tmin = -100.0; /* constants/cice/ice_constants_colpkg.F90:24-77, ice_therm_shared.F90:27 */

/* shift_ice :347-665.  Returns 0 or 1 (one of its four stops; nothing has been shifted then). */
static int shift_ice(int ntrcr, int ncat, const int *trcr_depend, int nt_Tsfc, double Tf, double *aicen, double *trcrn,
                     double *vicen, double *vsnon, double *hicen, const int *donor, double *daice, double *dvice)
{
    double atrcrn[MAXCAT + 1][MAXTR];
    for (int n = 1; n <= ncat; n++)
        for (int it = 0; it < ntrcr; it++) {
            const double b1 = trcr_depend[it] == 0 ? 1.0 : 0.0, b2 = trcr_depend[it] == 1 ? 1.0 : 0.0,
                         b3 = trcr_depend[it] == 2 ? 1.0 : 0.0;
            atrcrn[n][it] = trcrn[(size_t)(n - 1) * ntrcr + it] * (b1 * aicen[n] + b2 * vicen[n] + b3 * vsnon[n]);
        }
    for (int n = 1; n <= ncat - 1; n++) {
        int daice_negative = 0, dvice_negative = 0, daice_greater_aicen = 0, dvice_greater_vicen = 0, nd = 0, nr;
        if (donor[n] > 0) {
            nd = donor[n];
            if (daice[n] < c0) {
                if (daice[n] > -puny * aicen[nd]) {
                    daice[n] = c0;
                    dvice[n] = c0;
                } else
                    daice_negative = 1;
            }
            if (dvice[n] < c0) {
                if (dvice[n] > -puny * vicen[nd]) {
                    daice[n] = c0;
                    dvice[n] = c0;
                } else
                    dvice_negative = 1;
            }
            if (daice[n] > aicen[nd] * (c1 - puny)) {
                if (daice[n] < aicen[nd] * (c1 + puny)) {
                    daice[n] = aicen[nd];
                    dvice[n] = vicen[nd];
                } else
                    daice_greater_aicen = 1;
            }
            if (dvice[n] > vicen[nd] * (c1 - puny)) {
                if (dvice[n] < vicen[nd] * (c1 + puny)) {
                    daice[n] = aicen[nd];
                    dvice[n] = vicen[nd];
                } else
                    dvice_greater_vicen = 1;
            }
        }
        if (daice_negative && donor[n] > 0 && daice[n] <= -puny * aicen[nd]) return 1;
        if (dvice_negative && donor[n] > 0 && dvice[n] <= -puny * vicen[nd]) return 1;
        if (daice_greater_aicen && donor[n] > 0 && daice[n] >= aicen[donor[n]] * (c1 + puny)) return 1;
        if (dvice_greater_vicen && donor[n] > 0 && dvice[n] >= vicen[donor[n]] * (c1 + puny)) return 1;
        if (daice[n] > c0) {
            nd = donor[n];
            const double worka = daice[n] / aicen[nd];
            nr = nd == n ? nd + 1 : n;
            aicen[nd] = aicen[nd] - daice[n];
            aicen[nr] = aicen[nr] + daice[n];
            vicen[nd] = vicen[nd] - dvice[n];
            vicen[nr] = vicen[nr] + dvice[n];
            const double dvsnow = vsnon[nd] * worka;
            vsnon[nd] = vsnon[nd] - dvsnow;
            vsnon[nr] = vsnon[nr] + dvsnow;
            const double workb = dvsnow;
            for (int it = 0; it < ntrcr; it++) {
                const double b1 = trcr_depend[it] == 0 ? 1.0 : 0.0, b2 = trcr_depend[it] == 1 ? 1.0 : 0.0,
                             b3 = trcr_depend[it] == 2 ? 1.0 : 0.0;
                const double datrcr = trcrn[(size_t)(nd - 1) * ntrcr + it] * (b1 * daice[n] + b2 * dvice[n] + b3 * workb);
                atrcrn[nd][it] = atrcrn[nd][it] - datrcr;
                atrcrn[nr][it] = atrcrn[nr][it] + datrcr;
            }
        }
    }
    for (int n = 1; n <= ncat; n++) {
        if (aicen[n] > puny)
            hicen[n] = vicen[n] / aicen[n];
        else
            hicen[n] = c0;
        compute_tracers(ntrcr, trcr_depend, atrcrn[n], aicen[n], vicen[n], vsnon[n], nt_Tsfc, Tf,
                        trcrn + (size_t)(n - 1) * ntrcr);
    }
    return 0;
}

/* rebin :89-280.  A stop inside shift_ice is returned but, as in the reference, nobody looks at it: cleanup_itd goes
 * on to zap_small_areas, which clears the flag (:1102). */
static int rebin(int ntrcr, const int *trcr_depend, int nt_Tsfc, double Tf, double *aicen, double *trcrn, double *vicen,
                 double *vsnon, int ncat, const double *hin_max)
{
    int donor[MAXCAT + 1], stop = 0;
    double daice[MAXCAT + 1], dvice[MAXCAT + 1], hicen[MAXCAT + 1];
    for (int n = 1; n <= ncat; n++) {
        donor[n] = 0;
        daice[n] = c0;
        dvice[n] = c0;
        if (aicen[n] > puny)
            hicen[n] = vicen[n] / aicen[n];
        else
            hicen[n] = c0;
    }
    if (aicen[1] > puny) {
        if (hicen[1] <= hin_max[0] && hin_max[0] > c0) {
            aicen[1] = vicen[1] / hin_max[0];
            hicen[1] = hin_max[0];
        }
    }
    for (int n = 1; n <= ncat - 1; n++) { /* ice too thick for its category moves up */
        if (aicen[n] > puny && hicen[n] > hin_max[n]) {
            donor[n] = n;
            daice[n] = aicen[n];
            dvice[n] = vicen[n];
            stop = shift_ice(ntrcr, ncat, trcr_depend, nt_Tsfc, Tf, aicen, trcrn, vicen, vsnon, hicen, donor, daice, dvice);
            donor[n] = 0;
            daice[n] = c0;
            dvice[n] = c0;
        }
    }
    for (int n = ncat - 1; n >= 1; n--) { /* ice too thin moves down */
        if (aicen[n + 1] > puny && hicen[n + 1] <= hin_max[n]) {
            donor[n] = n + 1;
            daice[n] = aicen[n + 1];
            dvice[n] = vicen[n + 1];
            stop = shift_ice(ntrcr, ncat, trcr_depend, nt_Tsfc, Tf, aicen, trcrn, vicen, vsnon, hicen, donor, daice, dvice);
            donor[n] = 0;
            daice[n] = c0;
            dvice[n] = c0;
        }
    }
    return stop;
}

/* zap_snow :1340-1422 for one category */
static void zap_snow(double dt, int nslyr, double *trcrn, double *vsnon, double *dfresh, double *dfhocn, int nt_qsno)
{
    for (int k = 1; k <= nslyr; k++) {
        const double xtmp = trcrn[nt_qsno + k - 1] / dt * *vsnon / (double)nslyr;
        *dfhocn = *dfhocn + xtmp;
        trcrn[nt_qsno + k - 1] = c0;
    }
    const double xtmp = (rhos * *vsnon) / dt;
    *dfresh = *dfresh + xtmp;
    *vsnon = c0;
}

/* zap_small_areas :1029-1336.  Returns 0, 1 'Zap ice: negative ice area' or 2 'Zap ice: excess ice area'. */
static int zap_small_areas(double dt, double Tf, int ntrcr, int ncat, int nilyr, int nslyr, double *aice, double *aice0,
                           double *aicen, double *trcrn, double *vicen, double *vsnon, double *dfresh, double *dfsalt,
                           double *dfhocn, int *first_ice, int nt_Tsfc, int nt_qice, int nt_qsno)
{
    for (int n = 1; n <= ncat; n++) {
        double *tr = trcrn + (size_t)(n - 1) * ntrcr;
        if (aicen[n] < -puny)
            return 1;
        else if (fabs(aicen[n]) != c0 && fabs(aicen[n]) <= puny) {
            for (int k = 1; k <= nilyr; k++) {
                const double xtmp = tr[nt_qice + k - 1] / dt * vicen[n] / (double)nilyr;
                *dfhocn = *dfhocn + xtmp;
                tr[nt_qice + k - 1] = c0;
            }
            double xtmp = (rhoi * vicen[n]) / dt;
            *dfresh = *dfresh + xtmp;
            xtmp = rhoi * vicen[n] * ice_ref_salinity * p001 / dt;
            *dfsalt = *dfsalt + xtmp;
            *aice0 = *aice0 + aicen[n];
            aicen[n] = c0;
            vicen[n] = c0;
            tr[nt_Tsfc] = Tf;
            zap_snow(dt, nslyr, tr, &vsnon[n], dfresh, dfhocn, nt_qsno);
            for (int it = 1; it < ntrcr; it++) tr[it] = c0; /* tracers 2..ntrcr (:1203-1211), no brine */
            first_ice[n] = 1;
        }
    }
    if (*aice > (c1 + puny))
        return 2;
    else if (*aice > c1 && *aice < (c1 + puny)) {
        for (int n = 1; n <= ncat; n++) {
            const double *tr = trcrn + (size_t)(n - 1) * ntrcr;
            for (int k = 1; k <= nilyr; k++) {
                const double xtmp = tr[nt_qice + k - 1] * vicen[n] / (double)nilyr * (*aice - c1) / *aice / dt;
                *dfhocn = *dfhocn + xtmp;
            }
            for (int k = 1; k <= nslyr; k++) {
                const double xtmp = tr[nt_qsno + k - 1] * vsnon[n] / (double)nslyr * (*aice - c1) / *aice / dt;
                *dfhocn = *dfhocn + xtmp;
            }
            double xtmp = (rhoi * vicen[n] + rhos * vsnon[n]) * (*aice - c1) / *aice / dt;
            *dfresh = *dfresh + xtmp;
            xtmp = rhoi * vicen[n] * ice_ref_salinity * p001 * (*aice - c1) / *aice / dt;
            *dfsalt = *dfsalt + xtmp;
            aicen[n] = aicen[n] * (c1 / *aice);
            vicen[n] = vicen[n] * (c1 / *aice);
            vsnon[n] = vsnon[n] * (c1 / *aice);
        }
        *aice = c1;
        *aice0 = c0;
    }
    return 0;
}

/* zap_snow_temperature :1426-1559 (heat_capacity true) */
static void zap_snow_temperature(double dt, int ncat, int nslyr, int ntrcr, const double *aicen, double *trcrn, double *vsnon,
                                 double *dfresh, double *dfhocn, int nt_qsno)
{
    const double rnslyr = (double)nslyr;
    for (int n = 1; n <= ncat; n++) {
        double *tr = trcrn + (size_t)(n - 1) * ntrcr;
        int l_zap = 0;
        if (aicen[n] > puny) {
            const double hsn = vsnon[n] / aicen[n];
            for (int k = 1; k <= nslyr; k++) {
                double zqsn, Tmax;
                if (hsn > hs_min) {
                    zqsn = tr[nt_qsno + k - 1];
                    Tmax = -zqsn * puny * rnslyr / (rhos * cp_ice * vsnon[n]);
                } else {
                    zqsn = -rhos * Lfresh;
                    Tmax = puny;
                }
                const double zTsn = (Lfresh + zqsn / rhos) / cp_ice;
                if (zTsn < Tmin || zTsn > Tmax) l_zap = 1;
            }
        }
        if (l_zap) zap_snow(dt, nslyr, tr, &vsnon[n], dfresh, dfhocn, nt_qsno);
    }
}

/* cleanup_itd :754-1020 for one cell.  0-based category arrays of length ncat like ora_ridge_ice.  Returns 0 or 6:
 * colpkg_step_ridge replaces every stop of the clean-up by 'ice: ITD cleanup error in colpkg_step_ridge' (:3352-3354). */
int ora_cleanup_itd(double dt, double Tf, int ntrcr, int nilyr, int nslyr, int ncat, const double *hin_max, double *aicen0,
                    double *trcrn, double *vicen0, double *vsnon0, double *aice0, double *aice, const int *trcr_depend,
                    int nt_Tsfc, int nt_qice, int nt_qsno, int *first_ice0, double *fresh, double *fsalt, double *fhocn)
{
    double aicen[MAXCAT + 1], vicen[MAXCAT + 1], vsnon[MAXCAT + 1];
    int first_ice[MAXCAT + 1];
    if (ncat > MAXCAT || ntrcr > MAXTR) return -1;
    for (int n = 1; n <= ncat; n++) {
        aicen[n] = aicen0[n - 1];
        vicen[n] = vicen0[n - 1];
        vsnon[n] = vsnon0[n - 1];
        first_ice[n] = first_ice0[n - 1];
    }
    double dfresh = c0, dfsalt = c0, dfhocn = c0;
    int stop = 0;
    /* aggregate_area :53-81 */
    *aice = c0;
    for (int n = 1; n <= ncat; n++) *aice = *aice + aicen[n];
    *aice0 = fmax(c1 - *aice, c0);
    if (*aice > c1 + puny || *aice < -puny) return 6; /* 'aggregate ice area out of bounds' */
    if (*aice > puny) rebin(ntrcr, trcr_depend, nt_Tsfc, Tf, aicen, trcrn, vicen, vsnon, ncat, hin_max);
    stop = zap_small_areas(dt, Tf, ntrcr, ncat, nilyr, nslyr, aice, aice0, aicen, trcrn, vicen, vsnon, &dfresh, &dfsalt,
                           &dfhocn, first_ice, nt_Tsfc, nt_qice, nt_qsno);
    if (!stop) {
        zap_snow_temperature(dt, ncat, nslyr, ntrcr, aicen, trcrn, vsnon, &dfresh, &dfhocn, nt_qsno);
        *fresh = *fresh + dfresh;
        *fsalt = *fsalt + dfsalt;
        *fhocn = *fhocn + dfhocn;
    }
    for (int n = 1; n <= ncat; n++) {
        aicen0[n - 1] = aicen[n];
        vicen0[n - 1] = vicen[n];
        vsnon0[n - 1] = vsnon[n];
        first_ice0[n - 1] = first_ice[n];
    }
    return stop ? 6 : 0;
}

/* column_ridging's cell loop (mpas_seaice_column.F:3468-3530) over records: for cell c the category arrays are
 * aicen[c * ncat + n] and the tracers trcrn[(c * ncat + n) * ntrcr + it].  Per-category diagnostics likewise.
 * stop[c] receives the cell's return code; the first non-zero code is returned (the reference stops at that cell). */
int ora_ridge_cells(int nCells, double dt, int ndtd, int ncat, int nslyr, int ntrcr, double *hin_max, const double *rdg_conv,
                    const double *rdg_shear, const double *Tf, double *aicen, double *trcrn, double *vicen, double *vsnon,
                    double *aice0, const int *trcr_depend, int nt_Tsfc, int nt_qsno, int krdg_partic, int krdg_redist,
                    double mu_rdg, double *dardg1dt, double *dardg2dt, double *dvirdgdt, double *opening, double *fresh,
                    double *fhocn, double *aparticn, double *krdgn, double *aredistn, double *vredistn, double *dardg1ndt,
                    double *dardg2ndt, double *dvirdgndt, double *araftn, double *vraftn, int *stop, int *iterations,
                    int cleanup, int nilyr, int nt_qice, double *aice, double *fsalt, int *first_ice)
{
    int first = 0;
    for (int c = 0; c < nCells; c++) {
        const size_t k = (size_t)c * ncat;
        stop[c] = ora_ridge_ice(dt, ndtd, ncat, nslyr, ntrcr, hin_max, rdg_conv[c], rdg_shear[c], aicen + k,
                                trcrn + k * ntrcr, vicen + k, vsnon + k, aice0 + c, trcr_depend, nt_Tsfc, nt_qsno, krdg_partic,
                                krdg_redist, mu_rdg, dardg1dt + c, dardg2dt + c, dvirdgdt + c, opening + c, fresh + c,
                                fhocn + c, aparticn + k, krdgn + k, aredistn + k, vredistn + k, dardg1ndt + k, dardg2ndt + k,
                                dvirdgndt + k, araftn + k, vraftn + k, Tf[c], iterations + c);
        /* colpkg_step_ridge :3327-3354: the clean-up over the thermodynamic step dtt = dt * ndtd */
        if (!stop[c] && cleanup)
            stop[c] = ora_cleanup_itd(dt * ndtd, Tf[c], ntrcr, nilyr, nslyr, ncat, hin_max, aicen + k, trcrn + k * ntrcr,
                                      vicen + k, vsnon + k, aice0 + c, aice + c, trcr_depend, nt_Tsfc, nt_qice, nt_qsno,
                                      first_ice + k, fresh + c, fsalt + c, fhocn + c);
        if (stop[c] && !first) first = stop[c];
    }
    return first;
}

/* colpkg_aggregate, ice_colpkg.F90:3366-3480, for tracers without strata (trcr_base(it,:) the unit vector of
 * trcr_depend(it)), over records as above: trcr[c * ntrcr + it] receives the cell values.  This is what
 * seaice_column_aggregate (mpas_seaice_column.F:4391-4494) calls for every owned cell. */
void ora_aggregate_cells(int nCells, int ncat, int ntrcr, const double *Tf, const double *aicen, const double *trcrn,
                         const double *vicen, const double *vsnon, double *aice, double *trcr, double *vice, double *vsno,
                         double *aice0, const int *trcr_depend, int nt_Tsfc)
{
    double *atrcr = (double *)malloc(sizeof(double) * (size_t)(ntrcr > 0 ? ntrcr : 1));
    for (int c = 0; c < nCells; c++) {
        const size_t k = (size_t)c * ncat;
        double a = c0, v = c0, s = c0;
        for (int it = 0; it < ntrcr; it++) atrcr[it] = c0;
        for (int n = 0; n < ncat; n++) {
            a = a + aicen[k + n];
            v = v + vicen[k + n];
            s = s + vsnon[k + n];
            for (int it = 0; it < ntrcr; it++) {
                const double b1 = trcr_depend[it] == 0 ? c1 : c0, b2 = trcr_depend[it] == 1 ? c1 : c0,
                             b3 = trcr_depend[it] == 2 ? c1 : c0;
                const double atrcrn = trcrn[(k + n) * ntrcr + it] * (b1 * aicen[k + n] + b2 * vicen[k + n] + b3 * vsnon[k + n]);
                atrcr[it] = atrcr[it] + atrcrn;
            }
        }
        aice[c] = a;
        vice[c] = v;
        vsno[c] = s;
        aice0[c] = (c1 - a) > c0 ? (c1 - a) : c0;
        compute_tracers(ntrcr, trcr_depend, atrcr, a, v, s, nt_Tsfc, Tf[c], trcr + (size_t)c * ntrcr);
    }
    free(atrcr);
}
