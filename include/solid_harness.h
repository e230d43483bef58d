/*
 * solid_harness.h -- PETSc-free C-ABI harness reproducing the reference's host side of the path.
 *
 * The reference's host side is C on top of PETSc (`src/setuplibceed.c`, `src/matops.c`,
 * `src/misc.c`); PETSc is absent here, so this harness restates exactly those functions --
 * same names, same libCEED call sequences (cited per function in solid_harness.cpp) -- with
 * PETSc's DM replaced by the L-vector convention of this build: a "global"
 * vector is an L-vector whose Dirichlet entries are zero.  It uses nothing but include/ceed.h,
 * so the same source links against the MI355X backend (libsolid_harness_mi355x.so) and, for the
 * tests only, against the CPU oracle (oracle/libsolid_harness_oracle.so).
 */
#ifndef SOLID_HARNESS_H
#define SOLID_HARNESS_H

#include <ceed.h>

typedef struct SolidApp_private *SolidApp;

/* problemType of elasticity.h:40-61 (hyperFSIncomp is dead upstream code and not offered) */
typedef enum { ELAS_LIN = 0, ELAS_HYPER_SS = 1, ELAS_HYPER_FS = 2 } problemType;

/* Everything DMPlex hands the reference, as plain arrays (host):
 *   coords  [nvert][3]       vertex coordinates (DMGetCoordinatesLocal, setuplibceed.c:323-329)
 *   cells   [nelem][8]       vertex ids in tensor closure order (setupdm.c:194)
 *   for each level l < numLevels (coarse to fine, levelDegrees as cloptions.c:195-225):
 *     offsets[l] [nelem][P_l^3]  component-0 L-vector offset per node (CreateRestrictionPlex,
 *                                setuplibceed.c:194-240, sign of constrained dofs already dropped)
 *     lsizes[l]                  Ulocsz (elasticity.c:211-213)
 *     masks[l]   [lsizes[l]]     1 on constrained (Dirichlet) dofs, else 0
 */
CEED_EXTERN int SolidAppCreate(Ceed ceed, problemType problem, double nu, double E,
                               CeedInt numLevels, const CeedInt *levelDegrees, CeedInt qextra,
                               CeedInt nelem, CeedInt nvert, const CeedScalar *coords,
                               const CeedInt *cells, const CeedInt *const *offsets,
                               const CeedInt *lsizes, const unsigned char *const *masks,
                               SolidApp *app);
CEED_EXTERN int SolidAppDestroy(SolidApp *app);

/* src/matops.c, on CeedVectors in the L-layout of this build */
CEED_EXTERN int ApplyLocalCeedOp(SolidApp app, CeedOperator op, CeedVector X, CeedVector Y);      /* :26-60   */
CEED_EXTERN int FormResidual_Ceed(SolidApp app, CeedVector X, CeedVector Y);                    /* :63-79   */
CEED_EXTERN int ApplyJacobian_Ceed(SolidApp app, CeedInt level, CeedVector X, CeedVector Y);    /* :98-112  */
CEED_EXTERN int Prolong_Ceed(SolidApp app, CeedInt level, CeedVector Xc, CeedVector Yf);        /* :115-157 */
CEED_EXTERN int Restrict_Ceed(SolidApp app, CeedInt level, CeedVector Xf, CeedVector Yc);       /* :160-203 */
CEED_EXTERN int GetDiag_Ceed(SolidApp app, CeedInt level, CeedVector D);                        /* :206-244 */
/* -nu_smoother (matops.c:215-232): Poisson ratio used only while assembling the diagonal; < 0 clears */
CEED_EXTERN int SolidAppSetSmootherNu(SolidApp app, double nu_smoother);

/* Several GPUs: attach the interface sum of one level (CeedXHalo*, include/ceed.h).  Every function above then ends with
 * it, as the reference's end with DMLocalToGlobal(ADD_VALUES) (matops.c:57,153,199,238). */
CEED_EXTERN int SolidAppSetHalo(SolidApp app, CeedInt level, CeedXHalo halo);

/* accessors for tests / drivers */
CEED_EXTERN int SolidAppGetVectors(SolidApp app, CeedVector *qdata, CeedVector *gradu);
CEED_EXTERN int SolidAppGetLevelOperators(SolidApp app, CeedInt level, CeedOperator *opJacob,
                                          CeedOperator *opProlong, CeedOperator *opRestrict);
CEED_EXTERN int SolidAppGetResidualOperator(SolidApp app, CeedOperator *opApply);
CEED_EXTERN int SolidAppGetMultiplicityInverse(SolidApp app, CeedInt level, CeedVector *multinv);

#endif
