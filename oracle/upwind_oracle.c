/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's first-order upwind tracer
 * advection, src/shared/mpas_seaice_advection_upwind.F (config_advection_type = 'upwind').
 *
 * Nothing in the product path may call into this file: only tests/ use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no golden vectors for this path and its Fortran
 * cannot be built here without the MPAS framework modules.  The restatement is pinned by properties of
 * the scheme itself: exact conservation of the transported area on a closed domain, no change under zero
 * velocity, uniform fields stay uniform in a divergence-free flow.
 *
 * The loops follow the cited reference routines statement by statement, including the places where the
 * reference loops over all cells of a block rather than the owned ones.  Arrays use the reference's
 * Fortran layout (first index fastest, one spare trailing slot per mesh dimension) and 1-based
 * connectivity.  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stddef.h>

#define ICE_AREA_MINIMUM 1.0e-11 /* seaice_constants: iceAreaMinimum = seaicePuny */

/* edge_from_vertex_velocity, advection_upwind.F:1403-1535.  normalVectorEdge is (2, maxEdges, nCells+1). */
void upw_edge_from_vertex_velocity(int nCells, int nEdges, int maxEdges, const int *nEdgesOnCell,
                                   const int *edgesOnCell, const int *cellsOnEdge, const int *verticesOnEdge,
                                   const double *uVelocity, const double *vVelocity, const double *normalVectorEdge,
                                   double *edgeVelocity)
{
    for (int e = 0; e < nEdges; e++) edgeVelocity[e] = 0.0;
    for (int iCell = 1; iCell <= nCells; iCell++)
        for (int s = 1; s <= nEdgesOnCell[iCell - 1]; s++) {
            const int iEdge = edgesOnCell[(size_t)(iCell - 1) * maxEdges + (s - 1)];
            if (cellsOnEdge[(size_t)(iEdge - 1) * 2] == iCell) {
                double ue = 0.0, ve = 0.0;
                for (int k = 0; k < 2; k++) {
                    const int iVertex = verticesOnEdge[(size_t)(iEdge - 1) * 2 + k];
                    ue = ue + uVelocity[iVertex - 1];
                    ve = ve + vVelocity[iVertex - 1];
                }
                ue = ue / 2.0;
                ve = ve / 2.0;
                const double *n = normalVectorEdge + ((size_t)(iCell - 1) * maxEdges + (s - 1)) * 2;
                edgeVelocity[iEdge - 1] = ue * n[0] + ve * n[1];
            }
        }
}

/* prepare_none_parent_tracer, :819-915: the stand-in parent of the first tracer (1 wherever the cell or a
 * neighbour holds ice) and its edge flux (the edge velocity next to ice).  All arrays (1, nCategories, *). */
void upw_prepare_none_parent(int nCells, int nEdges, int nCategories, int maxEdges, const int *nEdgesOnCell,
                             const int *cellsOnCell, const int *cellsOnEdge, const double *childTracerOld,
                             const double *edgeVelocity, double *parentOld, double *parentNew, double *parentTend,
                             double *parentEdgeFlux)
{
    for (int iCell = 1; iCell <= nCells; iCell++)
        for (int k = 0; k < nCategories; k++) {
            const size_t i = (size_t)(iCell - 1) * nCategories + k;
            parentOld[i] = 0.0;
            parentNew[i] = 0.0;
            parentTend[i] = 0.0;
            if (childTracerOld[i] > ICE_AREA_MINIMUM) {
                parentOld[i] = 1.0;
                parentNew[i] = 1.0;
            }
            for (int s = 1; s <= nEdgesOnCell[iCell - 1]; s++) {
                const int cn = cellsOnCell[(size_t)(iCell - 1) * maxEdges + (s - 1)];
                if (childTracerOld[(size_t)(cn - 1) * nCategories + k] > ICE_AREA_MINIMUM) {
                    parentOld[i] = 1.0;
                    parentNew[i] = 1.0;
                    parentTend[i] = 0.0;
                    break;
                }
            }
        }
    for (int iEdge = 1; iEdge <= nEdges; iEdge++)
        for (int k = 0; k < nCategories; k++) {
            const int c1 = cellsOnEdge[(size_t)(iEdge - 1) * 2], c2 = cellsOnEdge[(size_t)(iEdge - 1) * 2 + 1];
            if (childTracerOld[(size_t)(c1 - 1) * nCategories + k] > ICE_AREA_MINIMUM ||
                childTracerOld[(size_t)(c2 - 1) * nCategories + k] > ICE_AREA_MINIMUM)
                parentEdgeFlux[(size_t)(iEdge - 1) * nCategories + k] = edgeVelocity[iEdge - 1];
            else
                parentEdgeFlux[(size_t)(iEdge - 1) * nCategories + k] = 0.0;
        }
}

/* upwind_tendencies, :1242-1389.  child arrays are (nTracers, nCategories, *); parentEdgeFlux and
 * parentTracerOld are the first layer of the parent's arrays, (nCategories, *). */
void upw_upwind_tendencies(int nCellsSolve, int nEdges, int nCategories, int nTracers, int maxEdges,
                           const int *nEdgesOnCell, const int *edgesOnCell, const int *cellsOnEdge,
                           const double *areaCell, const double *dvEdge, const int *interiorEdge,
                           const double *childTracerOld, const double *parentEdgeFlux, const double *parentTracerOld,
                           double parentTracerMinimum, double *childTendency, double *childEdgeFlux)
{
    for (int t = 0; t < nTracers; t++) {
        for (int iCell = 1; iCell <= nCellsSolve; iCell++)
            for (int k = 0; k < nCategories; k++) childTendency[((size_t)(iCell - 1) * nCategories + k) * nTracers + t] = 0.0;
        for (int iEdge = 1; iEdge <= nEdges; iEdge++)
            for (int k = 0; k < nCategories; k++) childEdgeFlux[((size_t)(iEdge - 1) * nCategories + k) * nTracers + t] = 0.0;
        for (int iCell = 1; iCell <= nCellsSolve; iCell++) {
            const double invAreaCell1 = 1.0 / areaCell[iCell - 1];
            for (int s = 1; s <= nEdgesOnCell[iCell - 1]; s++) {
                const int iEdge = edgesOnCell[(size_t)(iCell - 1) * maxEdges + (s - 1)];
                const int cell1 = cellsOnEdge[(size_t)(iEdge - 1) * 2], cell2 = cellsOnEdge[(size_t)(iEdge - 1) * 2 + 1];
                const int edgeSign = (iCell == cell1) ? -1 : 1;
                if (interiorEdge[iEdge - 1] != 1) continue;
                for (int k = 0; k < nCategories; k++) {
                    if (parentTracerOld[(size_t)(cell1 - 1) * nCategories + k] > parentTracerMinimum ||
                        parentTracerOld[(size_t)(cell2 - 1) * nCategories + k] > parentTracerMinimum) {
                        const double pf = parentEdgeFlux[(size_t)(iEdge - 1) * nCategories + k];
                        const double flux_upwind =
                            dvEdge[iEdge - 1] *
                            (fmax(0.0, pf) * childTracerOld[((size_t)(cell1 - 1) * nCategories + k) * nTracers + t] +
                             fmin(0.0, pf) * childTracerOld[((size_t)(cell2 - 1) * nCategories + k) * nTracers + t]);
                        const size_t ic = ((size_t)(iCell - 1) * nCategories + k) * nTracers + t;
                        childTendency[ic] = childTendency[ic] + edgeSign * flux_upwind * invAreaCell1;
                        childEdgeFlux[((size_t)(iEdge - 1) * nCategories + k) * nTracers + t] = flux_upwind / dvEdge[iEdge - 1];
                    }
                }
            }
        }
    }
}

/* the "calculate the new tracer" loop of run_advection_subvariable, :1198-1226 (all nCells of the block).
 * The reference indexes the parent with the child's layer in the second statement; the tracers it
 * connects all have one layer, which is what is restated here (nTracers == 1 for the parent access). */
void upw_new_tracer(int nCells, int nCategories, int nTracers, double *childTracerOld, double *childTracerNew,
                    const double *childTendency, const double *parentTracerOld, const double *parentTracerNew,
                    double parentTracerMinimum, double dt)
{
    for (int iCell = 1; iCell <= nCells; iCell++)
        for (int k = 0; k < nCategories; k++) {
            const size_t ip = (size_t)(iCell - 1) * nCategories + k;
            if (parentTracerNew[ip] > parentTracerMinimum)
                for (int t = 0; t < nTracers; t++) {
                    const size_t ic = ip * nTracers + t;
                    childTracerNew[ic] = childTracerOld[ic] * parentTracerOld[ip] + childTendency[ic] * dt;
                    childTracerOld[ic] = childTracerOld[ic] * parentTracerOld[ip];
                }
        }
}

/* scale_tracers_back_3D, :2199-2253 */
void upw_scale_back(int nCellsSolve, int nCategories, int nTracers, double *childTracerNew, const double *parentTracerNew)
{
    for (int iCell = 1; iCell <= nCellsSolve; iCell++)
        for (int k = 0; k < nCategories; k++)
            for (int t = 0; t < nTracers; t++) {
                const size_t ip = (size_t)(iCell - 1) * nCategories + k, ic = ip * nTracers + t;
                if (parentTracerNew[ip] > 0.0)
                    childTracerNew[ic] = childTracerNew[ic] / parentTracerNew[ip];
                else
                    childTracerNew[ic] = 0.0;
            }
}

/* volume <-> thickness, prepare_tracers :1949-1967 (toThickness, all cells) and finalize_tracers :2063-2081 */
void upw_convert(int nCellsLoop, int nCategories, int toThickness, const double *iceAreaCategory,
                 double *iceVolumeCategory, double *snowVolumeCategory)
{
    for (int iCell = 1; iCell <= nCellsLoop; iCell++)
        for (int k = 0; k < nCategories; k++) {
            const size_t i = (size_t)(iCell - 1) * nCategories + k;
            if (iceAreaCategory[i] > ICE_AREA_MINIMUM) {
                if (toThickness) {
                    iceVolumeCategory[i] = iceVolumeCategory[i] / iceAreaCategory[i];
                    snowVolumeCategory[i] = snowVolumeCategory[i] / iceAreaCategory[i];
                } else {
                    iceVolumeCategory[i] = iceVolumeCategory[i] * iceAreaCategory[i];
                    snowVolumeCategory[i] = snowVolumeCategory[i] * iceAreaCategory[i];
                }
            }
        }
}

/* one tracer of initial_conservation, :2492-2542: aggregate of child (* parent product) * area, and the
 * product left in the second time level (childParent).  parent == NULL for the 'none' parent. */
void upw_initial_conservation(int nCellsSolve, int nCategories, int nTracers, const double *areaCell,
                              const double *childTracer, const double *parentProduct, double *childParent,
                              double *tracerAggregate)
{
    for (int i = 0; i < nCategories * nTracers; i++) tracerAggregate[i] = 0.0;
    for (int iCell = 1; iCell <= nCellsSolve; iCell++)
        for (int k = 0; k < nCategories; k++)
            for (int t = 0; t < nTracers; t++) {
                const size_t ic = ((size_t)(iCell - 1) * nCategories + k) * nTracers + t;
                const size_t ia = (size_t)k * nTracers + t;
                if (!parentProduct) {
                    tracerAggregate[ia] = tracerAggregate[ia] + childTracer[ic] * areaCell[iCell - 1];
                    childParent[ic] = childTracer[ic];
                } else {
                    tracerAggregate[ia] = tracerAggregate[ia] + childTracer[ic] * parentProduct[ic] * areaCell[iCell - 1];
                    childParent[ic] = parentProduct[ic] * childTracer[ic];
                }
            }
}

/* one tracer of final_conservation, :2645-2656 */
void upw_final_conservation(int nCellsSolve, int nCategories, int nTracers, const double *areaCell,
                            const double *tracer, double *tracerAggregate)
{
    for (int i = 0; i < nCategories * nTracers; i++) tracerAggregate[i] = 0.0;
    for (int iCell = 1; iCell <= nCellsSolve; iCell++)
        for (int k = 0; k < nCategories; k++)
            for (int t = 0; t < nTracers; t++) {
                const size_t ia = (size_t)k * nTracers + t;
                tracerAggregate[ia] = tracerAggregate[ia] +
                                      tracer[((size_t)(iCell - 1) * nCategories + k) * nTracers + t] * areaCell[iCell - 1];
            }
}

/* interior_edges, mesh.F:567-614, for every edge of the block (the reference fills the owned edges and
 * takes the others from a halo exchange; the two agree on every edge of an owned cell). */
void upw_interior_edges(int nEdges, int nCells, const int *cellsOnEdge, int *interiorEdge)
{
    for (int iEdge = 1; iEdge <= nEdges; iEdge++) {
        interiorEdge[iEdge - 1] = 1;
        for (int k = 0; k < 2; k++)
            if (cellsOnEdge[(size_t)(iEdge - 1) * 2 + k] > nCells) interiorEdge[iEdge - 1] = 0;
    }
}
