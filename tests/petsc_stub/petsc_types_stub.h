/* TEST-ONLY stand-in for the PETSc headers: TYPE NAMES and the error macros, nothing else.  It exists so that
 * tests/test_boundary_compile.py can run `gcc -fsyntax-only` over the reference's own src/*.c against include/ceed.h and look at the
 * diagnostics on Ceed* / CEED_* identifiers.  No PETSc function is declared: calls to them are implicit
 * declarations, which the test ignores.  Nothing is ever compiled to code or linked against this. */
#ifndef PETSC_TYPES_STUB_H
#define PETSC_TYPES_STUB_H
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int PetscErrorCode;
typedef int PetscInt;
typedef int PetscMPIInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef double PetscLogDouble;
typedef int PetscLogStage;
typedef int PetscLogEvent;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
typedef int InsertMode;
typedef int ScatterMode;
typedef int NormType;
typedef int PetscCopyMode;
typedef int PetscDataType;
typedef int PetscMemType;
typedef int DMBoundaryType;
typedef int DMBoundaryConditionType;
typedef int PetscViewerFormat;
typedef int KSPConvergedReason;
typedef int SNESConvergedReason;
typedef int MatStructure;
typedef int MatOperation;
typedef int PCMGType;
typedef int PCMGCycleType;
typedef const char *VecType;
typedef const char *MatType;
typedef const char *DMType;
typedef const char *KSPType;
typedef const char *PCType;
typedef const char *SNESType;
typedef const char *SNESLineSearchType;
#define PETSC_STUB_OBJ(T) typedef struct _p_##T *T
PETSC_STUB_OBJ(Vec); PETSC_STUB_OBJ(Mat); PETSC_STUB_OBJ(DM); PETSC_STUB_OBJ(SNES); PETSC_STUB_OBJ(KSP); PETSC_STUB_OBJ(PC);
PETSC_STUB_OBJ(IS); PETSC_STUB_OBJ(PetscSection); PETSC_STUB_OBJ(DMLabel); PETSC_STUB_OBJ(PetscFE); PETSC_STUB_OBJ(PetscQuadrature);
PETSC_STUB_OBJ(PetscViewer); PETSC_STUB_OBJ(PetscDS); PETSC_STUB_OBJ(PetscSpace); PETSC_STUB_OBJ(PetscDualSpace); PETSC_STUB_OBJ(SNESLineSearch);
PETSC_STUB_OBJ(PetscSF); PETSC_STUB_OBJ(MatNullSpace); PETSC_STUB_OBJ(ISColoring); PETSC_STUB_OBJ(MatFDColoring); PETSC_STUB_OBJ(MatColoring);
PETSC_STUB_OBJ(PetscObject); PETSC_STUB_OBJ(PetscOptions); PETSC_STUB_OBJ(VecScatter); PETSC_STUB_OBJ(PetscPartitioner); PETSC_STUB_OBJ(DMField);
typedef void (*PetscVoidFunction)(void);
enum { INSERT_VALUES = 1, ADD_VALUES = 2, MAT_FINAL_ASSEMBLY = 0, FILE_MODE_WRITE = 1, MPI_IN_PLACE = 1, MPIU_REAL = 2, MPIU_SUM = 3, MPIU_SCALAR = 4,
       MPI_SUM = 5, MPIU_INT = 6, NORM_2 = 2, NORM_1 = 1, NORM_INFINITY = 3, PETSC_COPY_VALUES = 0, PETSC_OWN_POINTER = 1, PETSC_USE_POINTER = 2 };
enum { KSP_NORM_NATURAL = 3, MATOP_MULT = 3, MATOP_MULT_TRANSPOSE = 5, MATOP_GET_DIAGONAL = 17, MPI_DOUBLE = 7, MPI_MAX = 8, MPI_MIN = 9,
       PC_JACOBI_DIAGONAL = 0, PC_MG_CYCLE_V = 1, PC_MG_MULTIPLICATIVE = 0, DM_BC_ESSENTIAL = 1 };
typedef int PetscEnum;
#define KSPCG "cg"
#define KSPCHEBYSHEV "chebyshev"
#define KSPPREONLY "preonly"
#define MATAIJ "aij"
#define PCGAMG "gamg"
#define PCJACOBI "jacobi"
#define PCMG "mg"
#define SNESLINESEARCHCP "cp"
#define VECCUDA "cuda"
#define PETSCDUALSPACELAGRANGE "lagrange"
extern const char *const PCMGCycleTypes[], *const PCMGTypes[], *const *SNESConvergedReasons;
PetscErrorCode SNESComputeJacobianDefaultColor(SNES, Vec, Mat, Mat, void *);
#define PETSC_VERSION_LT(a, b, c) 0
#define PETSC_VERSION_GE(a, b, c) 1
/* the Vec array accessors the reference stores in function pointers (src/misc.c:58-66, 104-112) */
PetscErrorCode VecGetArray(Vec, PetscScalar **);
PetscErrorCode VecGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecRestoreArray(Vec, PetscScalar **);
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecCUDAGetArray(Vec, PetscScalar **);
PetscErrorCode VecCUDAGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecCUDARestoreArray(Vec, PetscScalar **);
PetscErrorCode VecCUDARestoreArrayRead(Vec, const PetscScalar **);
#define PetscFunctionBeginUser do { } while (0)
#define PetscFunctionBegin do { } while (0)
#define PetscFunctionReturn(x) return (x)
#define CHKERRQ(ierr) do { if (ierr) return (ierr); } while (0)
#define SETERRQ(...) return 1
#define SETERRQ1(...) return 1
#define SETERRQ2(...) return 1
#define SETERRQ3(...) return 1
#define PetscOptionsBegin(...) 0; do { } while (0)
#define PetscOptionsEnd() 0
#define PETSC_COMM_WORLD 0
#define PETSC_COMM_SELF 0
#define PETSC_DECIDE (-1)
#define PETSC_DETERMINE (-1)
#define PETSC_DEFAULT (-2)
#define PETSC_MAX_PATH_LEN 4096
#define PETSC_NULL NULL
#define PETSC_MACHINE_EPSILON 2.2e-16
#define PETSC_PI 3.14159265358979323846
#define PETSC_STATIC_INLINE static inline
#define PETSC_UNUSED
#endif
